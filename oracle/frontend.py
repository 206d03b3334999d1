"""TEST INFRASTRUCTURE - not a product path.  PARITY UNPINNED (see oracle/README.md).

The feature pipeline in front of the extractor:

    apply-cmvn-sliding --norm-vars=false --center=true --cmn-window=300 scp:feats.scp ark:- |
    select-voiced-frames ark:- scp,s,cs:vad.scp ark:- |
    (egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:79; the same string at train time,
     sid/nnet3/xvector/prepare_feats_for_egs.sh:66-71)

Both tools are upstream Kaldi binaries (not in /root/reference).  Restated from the published algorithm of
SlidingWindowCmn (feat/feature-functions.cc) [UPSTREAM, recalled]: for frame t the window is
[t - W/2, t - W/2 + W) when centred ([t - W, t + 1) otherwise, at least min_window frames), shifted to stay inside
[0, T) and clipped to it when T < W; the window mean (double accumulators) is subtracted; --norm-vars=false in every
recipe call.  select-voiced-frames keeps the rows whose VAD decision is non-zero.
"""
import numpy as np


def sliding_cmn(feats, cmn_window=300, center=True, min_window=100):
    feats = np.asarray(feats, dtype=np.float64)
    T = feats.shape[0]
    out = np.empty_like(feats)
    csum = np.concatenate([np.zeros((1, feats.shape[1])), np.cumsum(feats, axis=0)], axis=0)
    for t in range(T):
        if center:
            ws = t - cmn_window // 2
            we = ws + cmn_window
        else:
            ws = t - cmn_window
            we = t + 1
        if ws < 0:
            we -= ws
            ws = 0
        if not center:
            if we > t:
                we = max(t + 1, min_window)
        if we > T:
            ws -= (we - T)
            we = T
            if ws < 0:
                ws = 0
        mean = (csum[we] - csum[ws]) / (we - ws)
        out[t] = feats[t] - mean
    return out.astype(np.float32)


def select_voiced(feats, vad):
    vad = np.asarray(vad)
    if vad.shape[0] != feats.shape[0]:
        return None          # Kaldi: warning "mismatch in number of frames", utterance skipped
    keep = vad != 0
    if not keep.any():
        return None          # "no features were judged as voiced", skipped
    return np.asarray(feats)[keep]


def synthetic_vad(i, T, p_voiced=0.7):
    """Deterministic 0/1 decisions with speech-like runs (seeded per utterance)."""
    rng = np.random.default_rng(777 + i)
    v = np.zeros(T, dtype=np.float32)
    t = 0
    while t < T:
        run = int(rng.integers(5, 60))
        v[t:t + run] = 1.0 if rng.random() < p_voiced else 0.0
        t += run
    return v
