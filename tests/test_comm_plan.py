"""Region geometry of the one-shot all-reduce (host function of the extension, runs without
a GPU): the block regions of a call tile the message exactly once and fit the buffers."""
import os

import pytest

SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "eventgpt_amd", "_C.so")
pytestmark = pytest.mark.skipif(not os.path.exists(SO), reason="extension not built")


@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 17, 32, 64])
@pytest.mark.parametrize("n", [512, 1024, 1408, 4096, 5120, 8192, 11008])
def test_regions_tile_the_message(m, n):
    from eventgpt_amd.parallel.comm import plan
    got = plan(m, n)
    if got is None:
        assert m * ((n + 4095) // 4096) > 64 or n > 4096 * 8
        return
    blocks, chunk, bpr = got
    assert blocks == m * bpr and blocks <= 64 and chunk <= 4096 and chunk % 4 == 0
    covered = []
    for blk in range(blocks):                       # the kernel's indexing
        row, part = divmod(blk, bpr)
        e0 = row * n + part * chunk
        nq = min(chunk, n - part * chunk) // 4
        assert nq >= 0
        covered.append((e0, e0 + 4 * nq))
    covered.sort()
    pos = 0
    for a, b in covered:
        assert a == pos
        pos = b
    assert pos == m * n
    if m >= 4 and n <= 4096:
        assert bpr == 1                               # whole rows: the block can normalise them


def test_refuses_what_does_not_fit():
    from eventgpt_amd.parallel.comm import plan
    assert plan(65, 4096) is None
    assert plan(1, 6) is None and plan(0, 4096) is None
