import os

import numpy as np
import pytest
import torch

from eventgpt_amd.constants import EVENT_TOKEN_INDEX
from eventgpt_amd.data import (ByteTokenizer, Conversation, SeparatorStyle, conv_templates,
                               event_frames, load_event_npy, prepare_event_prompt, rasterize,
                               save_event_npy, split_by_count, split_by_time, synthetic_events,
                               tokenizer_event_token, voxel_grid)
from eventgpt_amd.data import clip_preprocess as cp
from eventgpt_amd.data.events import polarity_histogram, rasterize_sequential


def test_prompt_text_exact():
    p = prepare_event_prompt("What is in the scene?", "eventgpt_v1")
    assert p == ("A chat between a curious human and an artificial intelligence assistant. "
                 "The assistant gives helpful, detailed, and polite answers to the human's questions."
                 " USER: <ev_start><event><ev_end>\nWhat is in the scene? ASSISTANT:")


def test_conversation_styles():
    c = Conversation(system="S", roles=("A", "B"), messages=[["A", "hi"], ["B", "yo"], ["A", None]],
                     sep_style=SeparatorStyle.SINGLE, sep="###")
    assert c.get_prompt() == "S###A: hi###B: yo###A:"
    c.sep_style, c.sep, c.sep2 = SeparatorStyle.TWO, " ", "</s>"
    assert c.get_prompt() == "S A: hi B: yo</s>A:"
    c.sep_style = SeparatorStyle.PLAIN
    assert c.get_prompt() == "Shi yo</s>"
    c.sep_style, c.sep = SeparatorStyle.MPT, "<|im_end|>"
    assert c.get_prompt() == "S<|im_end|>Ahi<|im_end|>Byo<|im_end|>A"
    l2 = Conversation(system="sys", roles=("USER", "ASSISTANT"), messages=[["USER", "q1"], ["ASSISTANT", "a1"]],
                      sep_style=SeparatorStyle.LLAMA_2, sep="<s>", sep2="</s>")
    assert l2.get_prompt() == "[INST] <<SYS>>\nsys\n<</SYS>>\n\nq1 [/INST] a1 </s>"
    assert conv_templates["eventgpt_v1"].copy().messages == []


def test_tokenizer_event_token_single_bos():
    tok = ByteTokenizer(400)
    ids = tokenizer_event_token("ab<event>cd", tok)
    a, b, c, d = (ord(ch) + 3 for ch in "abcd")
    assert ids == [1, a, b, EVENT_TOKEN_INDEX, c, d]
    nb = ByteTokenizer(400, add_bos=False)
    assert tokenizer_event_token("ab<event>cd<event>", nb) == [a, b, EVENT_TOKEN_INDEX, c, d, EVENT_TOKEN_INDEX]
    t = tokenizer_event_token("x<event>", tok, return_tensors="pt")
    assert t.dtype == torch.long and t.tolist()[-1] == EVENT_TOKEN_INDEX


def test_byte_tokenizer_roundtrip_and_added_tokens():
    tok = ByteTokenizer(300)
    assert tok.add_tokens(["<ev_patch>"], special_tokens=True) == 1
    assert len(tok) == 301
    ids = tok("héllo <ev_patch>!").input_ids
    assert 300 in ids
    assert tok.decode(ids, skip_special_tokens=True) == "héllo !"
    assert tok.batch_decode(torch.tensor([ids]))[0].startswith("<s>héllo <ev_patch>!")


def test_split_by_count_remainder():
    ev = synthetic_events(1003, 64, 48)
    parts = split_by_count(ev, 5)
    assert [len(p["t"]) for p in parts] == [200, 200, 200, 200, 203]


def test_rasterize_matches_sequential_loop():
    ev = synthetic_events(4000, 40, 30, seed=3)
    for part in split_by_count(ev, 2):
        fast = rasterize(part["x"], part["y"], part["p"], conflict="last")
        slow = rasterize_sequential(part["x"], part["y"], part["p"])
        assert np.array_equal(fast, slow)


def test_rasterize_negative_wins_policy():
    x = np.array([1, 1, 2, 3]); y = np.array([0, 0, 0, 1]); p = np.array([0, 1, 1, 0])
    last = rasterize(x, y, p, conflict="last")
    neg = rasterize(x, y, p, conflict="negative")
    assert last.shape == (2, 4, 3)
    assert last[0, 1].tolist() == [255, 0, 0] and neg[0, 1].tolist() == [0, 0, 255]
    assert last[0, 2].tolist() == [255, 0, 0] and last[1, 3].tolist() == [0, 0, 255]
    assert last[1, 0].tolist() == [255, 255, 255]


def test_event_frames_canvas_is_data_dependent():
    ev = synthetic_events(5000, 64, 48)
    frames = event_frames(ev, 5)
    assert len(frames) == 5 and frames[0].shape == (48, 64, 3) and frames[0].dtype == np.uint8


def test_split_by_time():
    ev = {"x": np.zeros(6, int), "y": np.zeros(6, int), "p": np.zeros(6, int),
          "t": np.array([0, 10, 49_999, 50_000, 120_000, 149_999])}
    parts = split_by_time(ev, 50_000)
    assert [len(p["t"]) for p in parts] == [3, 1, 2]


def test_voxel_and_histogram():
    ev = synthetic_events(3000, 32, 24, seed=1)
    vol = voxel_grid(ev, bins=5)
    assert vol.shape == (5, 24, 32)
    pol = np.where(ev["p"] != 0, 1, -1).sum()
    assert vol.sum() == pytest.approx(pol)
    hist = polarity_histogram(ev, bins=5)
    assert hist.sum() == 3000 and hist.shape == (5, 2, 24, 32)
    assert np.array_equal(hist[:, 1].astype(np.float32) - hist[:, 0], vol)


def test_event_npy_roundtrip_safe_loader(tmp_path):
    ev = synthetic_events(1000, 64, 48)
    p = str(tmp_path / "s.npy")
    save_event_npy(p, ev)                      # pickled dict, the sample-file format
    back = load_event_npy(p)                   # symbolic reader: nothing in the file is executed
    for k in "xytp":
        assert np.array_equal(back[k], ev[k]) and back[k].dtype == ev[k].dtype
    pz = str(tmp_path / "s.npz")
    save_event_npy(pz, ev)
    assert np.array_equal(load_event_npy(pz)["t"], ev["t"])


def test_safe_loader_never_executes_the_file(tmp_path, capsys):
    """A pickle that would call a function on load is read as a tree of names and refused:
    the callable is never looked up, let alone run."""
    ran = []

    class Evil:
        def __reduce__(self):
            return (ran.append, ("pwned",))
    p = str(tmp_path / "evil.npy")
    np.save(p, np.array({"x": Evil()}, dtype=object), allow_pickle=True)
    with pytest.raises((ValueError, KeyError)):
        load_event_npy(p)
    assert ran == []

    class Sneaky:                       # a global outside numpy's array protocol in place of an array
        def __reduce__(self):
            return (os.getcwd, ())
    np.save(p, np.array({k: Sneaky() for k in "xytp"}, dtype=object), allow_pickle=True)
    with pytest.raises(ValueError):
        load_event_npy(p)


def test_resize_matches_pil_bicubic():
    from PIL import Image
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    for (h, w) in [(33, 44), (24, 32), (96, 128), (48, 30)]:
        ours = cp.resize_bicubic_u8(img, h, w)
        pil = np.asarray(Image.fromarray(img).resize((w, h), resample=Image.BICUBIC))
        assert np.array_equal(ours, pil), (h, w, np.abs(ours.astype(int) - pil).max())


def test_preprocess_matches_hf_clip_processor():
    transformers = pytest.importorskip("transformers")
    from transformers import CLIPImageProcessor
    proc = CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336},
                              do_resize=True, do_center_crop=True, do_normalize=True, resample=3,
                              image_mean=[0.48145466, 0.4578275, 0.40821073],
                              image_std=[0.26862954, 0.26130258, 0.27577711])
    frame = event_frames(synthetic_events(20000, 640, 480), 1)[0]
    want = proc(frame, return_tensors="pt")["pixel_values"][0]
    got = torch.from_numpy(cp.preprocess_image(frame))
    assert got.shape == (3, 336, 336)
    assert (got - want).abs().max().item() < 1e-5


def test_resample_table_vectorised_builder_equals_per_coordinate_form():
    """The table builder that works on all output coordinates at once gives the same integer taps as
    the one-coordinate-at-a-time specification (down- and up-scaling, tiny and odd sizes)."""
    import random

    import numpy as np

    from eventgpt_amd.data import clip_preprocess as cp
    rnd = random.Random(3)
    sizes = [(640, 448), (480, 336), (336, 336), (346, 336), (260, 336), (1280, 597), (100, 336), (5, 336),
             (337, 336)] + [(rnd.randint(20, 1500), rnd.randint(300, 700)) for _ in range(12)]
    for n_in, n_out in sizes:
        want = cp._resample_table_loop(n_in, n_out)
        got = cp.resample_table(n_in, n_out)
        assert np.array_equal(want.start, got.start) and np.array_equal(want.count, got.count), (n_in, n_out)
        assert np.array_equal(want.weight, got.weight), (n_in, n_out)
