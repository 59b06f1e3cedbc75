"""Host-side cache bound of the serving engine (the logic itself needs no GPU)."""
import types

import torch


def _stub(l_max, pos):
    from eventgpt_amd.runtime.engine import GenerationEngine
    s = types.SimpleNamespace(l_max=l_max, pos=torch.tensor([pos], dtype=torch.int32), _pos_hi=0)
    s.cache_room = types.MethodType(GenerationEngine.cache_room, s)
    return s


def test_bound_is_an_upper_bound_and_resyncs_when_it_says_full():
    s = _stub(100, 40)
    s._pos_hi = 40
    assert s.cache_room(6)                  # 46 <= 100: no device read needed
    s._pos_hi = 97                          # over-estimated by earlier verifies (actual position 40)
    assert s.cache_room(6)                  # says full -> re-reads the exact position -> fits
    assert s._pos_hi == 40
    s.pos.fill_(95)
    s._pos_hi = 95
    assert not s.cache_room(6)              # 101 > 100: refused
    assert s.cache_room(5)


def test_verify_advances_the_bound_by_the_most_it_can_append():
    s = _stub(64, 58)
    s._pos_hi = 58
    assert s.cache_room(6)
    s._pos_hi += 6
    s.pos.fill_(60)                         # only 2 rows were really kept
    assert not s.cache_room(6) and s._pos_hi == 60
