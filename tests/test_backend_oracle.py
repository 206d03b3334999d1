"""CPU tests of the speaker-level back-end restatement (oracle/backend.py) against hand-computed known answers —
the tools it follows are upstream Kaldi binaries called at egs/sre/v2/run_sre10.sh:219-241 and
sid/nnet3/xvector/extract_xvectors_new.sh:106-107 (parity unpinned, see oracle/README.md)."""
import numpy as np
import pytest

from oracle import backend as B


def test_speaker_means_known_answer():
    vec = {"a1": np.array([1, 2, 3], np.float32), "a2": np.array([3, 2, 1], np.float32), "b1": np.array([4, 4, 4], np.float32)}
    spk2utt = [("A", ["a1", "a2", "a3"]), ("B", ["b1"]), ("C", ["c1"])]
    means, counts, missing, empty = B.speaker_means(spk2utt, vec)
    assert [k for k, _ in means] == ["A", "B"]                  # spk2utt order, speakers without vectors dropped
    np.testing.assert_array_equal(means[0][1], np.array([2, 2, 2], np.float32))
    np.testing.assert_array_equal(means[1][1], np.array([4, 4, 4], np.float32))
    assert counts == {"A": 2, "B": 1} and missing == ["a3", "c1"] and empty == ["C"]


def test_speaker_mean_is_fp32_sequential():
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((37, 5)) * 1e3).astype(np.float32)
    vec = {"u%d" % i: x[i] for i in range(37)}
    (_, m), = B.speaker_means([("S", list(vec))], vec)[0]
    acc = x[0].copy()
    for i in range(1, 37):
        acc = (acc + x[i]).astype(np.float32)
    np.testing.assert_array_equal(m, acc * np.float32(1.0 / 37))


def test_global_mean_and_subtraction():
    x = np.array([[1, 10], [3, 30], [8, 20]], np.float32)
    np.testing.assert_allclose(B.global_mean(x), [4, 20])
    np.testing.assert_allclose(B.subtract_global_mean(x), [[-3, -10], [-1, 10], [4, 0]])
    np.testing.assert_allclose(B.subtract_global_mean(x, [1, 1]), x - 1)


def test_transform_vec_linear_affine_and_mismatch():
    x = np.array([[1, 2, 3]], np.float32)
    lin = np.array([[1, 0, 0], [0, 1, 1]], np.float32)
    aff = np.array([[1, 0, 0, 5], [0, 1, 1, -1]], np.float32)
    np.testing.assert_allclose(B.transform_vec(x, lin), [[1, 5]])
    np.testing.assert_allclose(B.transform_vec(x, aff), [[6, 4]])
    with pytest.raises(ValueError, match="Dimension mismatch"):
        B.transform_vec(x, np.zeros((2, 5), np.float32))


def test_normalize_length_known_answers():
    x = np.array([[3, 4], [0, 0], [1, 0]], np.float32)
    y, r = B.normalize_length(x)
    np.testing.assert_allclose(r, [5 / np.sqrt(2), 0, 1 / np.sqrt(2)])
    np.testing.assert_allclose(np.linalg.norm(y[0]), np.sqrt(2), rtol=1e-6)       # length sqrt(dim)
    np.testing.assert_array_equal(y[1], [0, 0])                                   # zero vector left alone
    y2, r2 = B.normalize_length(x, scaleup=False)
    np.testing.assert_allclose(y2[0], [0.6, 0.8], rtol=1e-6)
    np.testing.assert_allclose(r2, [5, 0, 1])
    y3, _ = B.normalize_length(x, normalize=False)
    np.testing.assert_array_equal(y3, x)


def test_chain_equals_the_stages():
    rng = np.random.default_rng(11)
    x = rng.standard_normal((9, 12)).astype(np.float32)
    mean = rng.standard_normal(12).astype(np.float32)
    t = rng.standard_normal((5, 12)).astype(np.float32)
    staged, _ = B.normalize_length(B.transform_vec(B.subtract_global_mean(x, mean), t))
    fused, _ = B.backend_chain(x, mean, t, normalize=True)
    np.testing.assert_allclose(fused, staged, rtol=2e-6, atol=2e-6)
