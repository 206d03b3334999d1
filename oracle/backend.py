"""TEST INFRASTRUCTURE - not a product path.  PARITY UNPINNED (see oracle/README.md).

The vector tools the reference runs on the extracted embeddings:

    ivector-mean ark:$data/spk2utt scp:xvector.scp ark,scp:spk_xvector.ark,spk_xvector.scp ark,t:num_utts.ark
        (egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:106-107)
    ivector-mean scp:xvector.scp mean.vec                                            (egs/sre/v2/run_sre10.sh:219-221)
    ivector-subtract-global-mean [mean.vec] ... | transform-vec transform.mat ... | ivector-normalize-length ...
        (run_sre10.sh:229,233,238-241)

All four are upstream Kaldi binaries (ivectorbin/, bin/transform-vec.cc; not in /root/reference).  Restated from their
published behaviour [UPSTREAM, recalled]:
  * ivector-mean, speaker form: for each spk2utt line, the vectors of the utterances present in the table are added in
    list order into a BaseFloat vector and scaled by 1/count; utterances without a vector are warned about; a speaker
    with none produces no output; num_utts gets the count.  Global form: double accumulator over the whole table.
  * ivector-subtract-global-mean: x - mean (mean read from a file, or the mean of the input itself).
  * transform-vec: y = M x when M has dim columns, y = M[:, :dim] x + M[:, dim] when it has dim + 1, error otherwise.
  * ivector-normalize-length: ratio = |x| / sqrt(dim) (|x| with --scaleup=false); x /= ratio unless ratio == 0
    ("Zero iVector" warning) or --normalize=false.
"""
import numpy as np


def speaker_means(spk2utt, vectors):
    """spk2utt: [(spk, [utt, ...])], vectors: {utt: float32 vector}.  Returns ([(spk, mean float32)], {spk: count},
    missing utterances, speakers without output) — fp32 accumulation in list order, like the AddVec loop."""
    out, counts, missing, empty = [], {}, [], []
    for spk, utts in spk2utt:
        acc, n = None, 0
        for u in utts:
            if u not in vectors:
                missing.append(u)
                continue
            v = np.asarray(vectors[u], dtype=np.float32)
            acc = v.copy() if acc is None else (acc + v).astype(np.float32)
            n += 1
        if n == 0:
            empty.append(spk)
            continue
        out.append((spk, (acc * np.float32(1.0 / n)).astype(np.float32)))
        counts[spk] = n
    return out, counts, missing, empty


def global_mean(vectors):
    """vectors: [n, dim] -> float32 mean with a double accumulator."""
    x = np.asarray(vectors, dtype=np.float32)
    acc = np.zeros(x.shape[1], dtype=np.float64)
    for row in x:
        acc += row
    return (acc * (1.0 / x.shape[0])).astype(np.float32)


def subtract_global_mean(x, mean=None):
    x = np.asarray(x, dtype=np.float32)
    if mean is None:
        mean = global_mean(x)
    return (x - np.asarray(mean, dtype=np.float32)).astype(np.float32)


def transform_vec(x, m):
    x = np.asarray(x, dtype=np.float64)
    m = np.asarray(m, dtype=np.float64)
    dim = x.shape[1]
    if m.shape[1] == dim:
        return (x @ m.T).astype(np.float32)
    if m.shape[1] == dim + 1:
        return (x @ m[:, :dim].T + m[:, dim]).astype(np.float32)
    raise ValueError("Dimension mismatch: input vector has dimension %d and transform has %d columns" % (dim, m.shape[1]))


def normalize_length(x, normalize=True, scaleup=True):
    """Returns (vectors, ratios)."""
    x = np.asarray(x, dtype=np.float32)
    out = x.copy()
    ratios = np.zeros(x.shape[0], dtype=np.float64)
    for i, v in enumerate(x):
        norm = float(np.sqrt(np.sum(v.astype(np.float64) ** 2)))
        ratio = norm / np.sqrt(x.shape[1]) if scaleup else norm
        ratios[i] = ratio
        if ratio != 0.0 and normalize:
            out[i] = (v.astype(np.float64) / ratio).astype(np.float32)
    return out, ratios


def backend_chain(x, mean=None, transform=None, normalize=False, scaleup=True):
    """subtract mean -> transform -> normalise, the chain of run_sre10.sh:239-240, in float64 between the stages."""
    y = np.asarray(x, dtype=np.float64)
    if mean is not None:
        y = y - np.asarray(mean, dtype=np.float64)
    if transform is not None:
        m = np.asarray(transform, dtype=np.float64)
        dim = y.shape[1]
        if m.shape[1] == dim:
            y = y @ m.T
        elif m.shape[1] == dim + 1:
            y = y @ m[:, :dim].T + m[:, dim]
        else:
            raise ValueError("Dimension mismatch")
    ratios = np.sqrt(np.sum(y * y, axis=1))
    if scaleup:
        ratios = ratios / np.sqrt(y.shape[1])
    if normalize:
        nz = ratios != 0
        y[nz] = y[nz] / ratios[nz, None]
    return y.astype(np.float32), ratios
