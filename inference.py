"""Drop-in entry point with the reference's flags (reference: inference.py:11-66).

    python inference.py --model_path ./checkpoints/EventGPT-7b \
        --event_frame ./samples/sample1.npy --query "Describe the scene."
"""
from eventgpt_amd.cli.inference import run

if __name__ == "__main__":
    run()
