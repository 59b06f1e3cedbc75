#!/bin/bash
# The research harness end to end at the 7B geometry (random weights, synthetic data):
# extract paired hidden states -> dump the verifier LM head -> train an adapter ->
# offline acceptance metrics -> LoRA distillation steps.  Prints wall time per stage.
# Usage: bash benchmarks/harness_chain.sh [work_dir] [n_scenes]
set -e
W=${1:-/tmp/evgpt_chain}
N=${2:-8}
rm -rf "$W"; mkdir -p "$W"
stage() { local name=$1; shift; local t0=$(date +%s%N); "$@" > "$W/$name.log" 2>&1 || { tail -20 "$W/$name.log"; exit 1; }; local t1=$(date +%s%N); echo "$name: $(( (t1 - t0) / 1000000 )) ms"; }
python - <<PY
from eventgpt_amd.pipeline.dataset import make_synthetic_dataset
make_synthetic_dataset("$W/ds", n_samples=$N, n_events=60000, width=640, height=480)
PY
stage extract python -m eventgpt_amd.pipeline.extract_hidden_states --random_init 7b --dataset_dir "$W/ds" --output_dir "$W/hs" --max_questions 2 --max_new_tokens 30 --chunk_size 8
STORE=$(ls -d "$W"/hs/chunked_* | head -1)
stage lm_head python -m eventgpt_amd.pipeline.extract_vl_lm_head --random_init 7b --output "$W/vl_lm_head.pt"
stage train_L1 python -m eventgpt_amd.align.trainer --train_data "$STORE" --adapter_level 1 --num_epochs 3 --batch_size 8 --output_dir "$W/tasks"
CK=$(ls "$W"/tasks/L1/*/best_model.pt | head -1)
stage evaluate python -m eventgpt_amd.eval.acceptance --checkpoint "$CK" --test_data "$STORE" --lm_head "$W/vl_lm_head.pt" --output_dir "$W/eval" --no_plots
stage train_lora python -m eventgpt_amd.align.train_lora --random_init 7b --existing_chunks "$STORE" --dataset_dir "$W/ds" --max_questions 2 --num_epochs 1 --gradient_accumulation 4 --output_dir "$W/lora" --event_image_key event_image
python - <<PY
import json
m = json.load(open("$W/eval/acceptance_metrics.json"))
print("pairs", m.get("batch_size"), "cos_sim_mean", round(m["cos_sim_mean"], 4), "token_match", round(m.get("token_match_rate", float("nan")), 4))
h = json.load(open("$W/lora/history.json"))
print("lora history", h if isinstance(h, list) else {k: h[k] for k in list(h)[:4]})
PY
