"""Seeded synthetic inputs shared by bench.py, smoke() and the tests."""
import numpy as np


def utterance(T: int, utt: int, dur_mode: str = "const7", hidden: int = 528):
    """phoneme ids ~ U{0..27}, punct ids ~ U{0..9} from default_rng(1234+utt); unit-norm N(0,1) speaker
    embedding; forced durations: 'const7' (=7 each), 'uniform' (U{3..10}) or None (predicted)."""
    r = np.random.default_rng(1234 + utt)
    phoneme = r.integers(0, 28, size=T).astype(np.int32)
    puncts = r.integers(0, 10, size=T).astype(np.int32)
    spk = r.standard_normal(hidden)
    spk = (spk / np.linalg.norm(spk)).astype(np.float32)
    if dur_mode == "const7":
        dur = np.full(T, 7, dtype=np.int32)
    elif dur_mode == "uniform":
        dur = r.integers(3, 11, size=T).astype(np.int32)
    else:
        dur = None
    return phoneme, puncts, spk, dur


def batch(B: int, T: int, first_utt: int = 0, dur_mode: str = "const7", hidden: int = 528):
    ph = np.zeros((B, T), np.int32); pu = np.zeros((B, T), np.int32)
    spk = np.zeros((B, hidden), np.float32)
    dur = np.zeros((B, T), np.int32) if dur_mode else None
    for b in range(B):
        p, q, s, d = utterance(T, first_utt + b, dur_mode, hidden)
        ph[b], pu[b], spk[b] = p, q, s
        if dur is not None:
            dur[b] = d
    return ph, pu, np.full(B, T, np.int32), spk, dur
