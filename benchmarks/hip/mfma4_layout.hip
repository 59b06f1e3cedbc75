// Lane layout of v_mfma_f32_4x4x4_16b_bf16 (16 independent 4x4x4 products per wave), checked against a
// host model: A of block b, row i, k in lane 4 b + i element k; B[k][j] in lane 4 b + j element k;
// D[i][j] in lane 4 b + j register i.   hipcc --offload-arch=gfx950 -O2 mfma4_layout.hip -o mfma4_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(4))) float f4;
static unsigned short bf(float x) { unsigned int u; std::memcpy(&u, &x, 4); return (unsigned short)(u >> 16); }
__global__ void k(const s4* a, const s4* b, f4* c) {
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  c[threadIdx.x] = acc;
}
int main() {
  float A[16][4][4], B[16][4][4];   // [block][i][k], [block][k][j]
  for (int b = 0; b < 16; ++b) for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) {
    A[b][x][y] = (float)((b * 7 + x * 3 + y) % 13 - 6);
    B[b][x][y] = (float)((b * 5 + x * 2 + y * 11) % 9 - 4);
  }
  unsigned short ha[64][4], hb[64][4];
  for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
    ha[l][e] = bf(A[l / 4][l % 4][e]);       // row i = l % 4, k = e
    hb[l][e] = bf(B[l / 4][e][l % 4]);       // k = e, column j = l % 4
  }
  s4 *da, *db; f4* dc; float hc[64][4];
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dc, sizeof(hc));
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc);
  hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const int b = l / 4, j = l % 4, i = r;
    float want = 0; for (int kk = 0; kk < 4; ++kk) want += A[b][i][kk] * B[b][kk][j];
    if (std::fabs(want - hc[l][r]) > 1e-3f) { if (bad < 8) printf("lane %d reg %d: got %g want %g\n", l, r, hc[l][r], want); ++bad; }
  }
  printf(bad ? "MISMATCH %d\n" : "layout ok\n", bad);
  return bad != 0;
}
