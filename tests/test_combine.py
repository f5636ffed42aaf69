"""Weighted combination of sub-embeddings: the oracle against literal numpy, host argument checks,
and (gpu) the mq_weighted_combine kernel against the oracle."""
import numpy as np
import pytest
import torch

from oracle import combine as OC


def test_oracle_matches_literal_reference_expressions():
    rng = np.random.default_rng(0)
    vecs = [rng.standard_normal(16).tolist() for _ in range(3)]
    w = [0.7, -0.2, 1.5]
    # tensor_fields_container.py:356-363
    combo = [np.array(v) * wt for v, wt in zip(vecs, w)]
    lit = np.squeeze(np.mean(combo, axis=0))
    assert np.array_equal(OC.combine_multimodal(vecs, w, False), lit)
    assert np.array_equal(OC.combine_multimodal(vecs, w, True), lit / np.linalg.norm(lit))
    # tensor_search.py:1954-1962
    merged = np.mean([np.asarray(v) * wt for v, wt in zip(vecs, w)], axis=0)
    assert np.array_equal(OC.combine_query(vecs, w, False), merged)
    assert np.allclose(np.linalg.norm(OC.combine_query(vecs, w, True)), 1.0)
    # zero vector: the search side leaves it, the add-documents side divides by zero -> NaN
    z = [[0.0] * 4, [0.0] * 4]
    assert np.array_equal(OC.combine_query(z, [1, 1], True), np.zeros(4))
    assert np.isnan(OC.combine_multimodal(z, [1, 1], True)).all()


def test_combine_needs_gpu_and_validates():
    from marqo_amd import combine as MC
    from marqo_amd._lib import MarqoHipUnavailableError
    with pytest.raises(MarqoHipUnavailableError):
        MC.combine_weighted(np.zeros((2, 4), np.float32), [[(0, 1.0)]], MC.RAW, device="cpu")


@pytest.mark.gpu
def test_kernel_matches_oracle():
    from marqo_amd import combine as MC
    rng = np.random.default_rng(1)
    for D in (4, 512, 768, 1000, 2048):
        n = 37
        emb = rng.standard_normal((n, D)).astype(np.float32)
        groups, k = [], 0
        sizes = [1, 2, 3, 10, 1, 5]
        for s in sizes:
            groups.append([(int(rng.integers(0, n)), float(rng.uniform(-2, 2))) for _ in range(s)])
        for mode, fn, norm in ((MC.RAW, OC.combine_query, False), (MC.NORMALIZE, OC.combine_multimodal, True),
                               (MC.NORMALIZE_IF_NONZERO, OC.combine_query, True)):
            out = MC.combine_weighted(emb, groups, mode).cpu().numpy()
            for g, terms in enumerate(groups):
                ref = fn([emb[r] for r, _ in terms], [np.float32(w) for _, w in terms], norm)
                np.testing.assert_allclose(out[g], ref.astype(np.float32), rtol=2e-6, atol=1e-7)


@pytest.mark.gpu
def test_kernel_edge_cases_and_batch_wrappers():
    from marqo_amd import combine as MC
    emb = np.zeros((3, 8), np.float32)
    emb[2] = np.arange(8)
    out = MC.combine_weighted(emb, [[(0, 1.0), (1, 2.0)], [(2, 0.5)]], MC.NORMALIZE_IF_NONZERO).cpu().numpy()
    assert np.array_equal(out[0], np.zeros(8))
    np.testing.assert_allclose(out[1], emb[2] / np.linalg.norm(emb[2]), rtol=1e-6)
    out = MC.combine_weighted(emb, [[(0, 1.0), (1, 2.0)]], MC.NORMALIZE).cpu().numpy()
    assert np.isnan(out).all()
    assert MC.combine_weighted(emb, [], MC.RAW).shape == (0, 8)
    with pytest.raises(IndexError):
        MC.combine_weighted(emb, [[(3, 1.0)]], MC.RAW)
    with pytest.raises(ValueError):
        MC.combine_weighted(emb, [[]], MC.RAW)
    # tensor rows given as a device tensor with a row stride
    dev = torch.from_numpy(np.random.default_rng(2).standard_normal((5, 16)).astype(np.float32)).cuda()
    out = MC.combine_weighted(dev[:, :8], [[(0, 1.0), (4, -1.0)]], MC.RAW).cpu().numpy()
    np.testing.assert_allclose(out[0], ((dev[0, :8] - dev[4, :8]) / 2).cpu().numpy(), rtol=1e-6, atol=1e-7)
    # batch wrappers mirror the two reference call sites
    rng = np.random.default_rng(3)
    docs = [{"title": rng.standard_normal(12).tolist(), "image": rng.standard_normal(12).tolist()},
            {"image": rng.standard_normal(12).tolist()}]
    weights = {"title": 0.3, "image": 0.7, "absent": 1.0}
    got = MC.combine_multimodal_fields(docs, weights, normalize=True)
    for d, doc in enumerate(docs):
        names = [k for k in weights if k in doc]
        ref = OC.combine_multimodal([np.float32(doc[k]) for k in names], [np.float32(weights[k]) for k in names], True)
        np.testing.assert_allclose(got[d], ref.astype(np.float32), rtol=2e-6, atol=1e-7)
    q = [[(rng.standard_normal(12).tolist(), 1.0), (rng.standard_normal(12).tolist(), -0.5)], [(np.zeros(12).tolist(), 2.0)]]
    got = MC.combine_query_vectors(q, normalize=True)
    for i, terms in enumerate(q):
        ref = OC.combine_query([np.float32(v) for v, _ in terms], [np.float32(w) for _, w in terms], True)
        np.testing.assert_allclose(got[i], ref.astype(np.float32), rtol=2e-6, atol=1e-7)
