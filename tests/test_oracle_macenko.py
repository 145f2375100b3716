"""oracle/macenko.py has no reference to be pinned to (the reference does not contain Macenko normalisation: parity unpinned).  What CAN be
checked is the algorithm against its own definition: tiles synthesised from known stain vectors give those vectors back, and
normalising maps every stain basis onto the reference basis."""
import numpy as np

from oracle import macenko as om


def _ang(a, b):
    return float(np.degrees(np.arccos(np.clip(abs(a @ b) / np.linalg.norm(a) / np.linalg.norm(b), 0, 1))))


def test_fit_recovers_the_stain_vectors_the_tile_was_made_from():
    he = np.array([[0.65, 0.07], [0.70, 0.99], [0.29, 0.11]])
    he /= np.linalg.norm(he, axis=0, keepdims=True)
    for t in om.synthetic_he_tiles(3, 160, seed=4, he=he):
        fit, maxc = om.macenko_fit(t)
        # haematoxylin to ~1 degree; eosin is biased towards it by design: beta rejects the purest eosin pixels (their red OD is below 0.15)
        assert _ang(fit[:, 0], he[:, 0]) < 2.5 and _ang(fit[:, 1], he[:, 1]) < 9.0
        assert fit[0, 0] > fit[0, 1] and (maxc > 0).all()


def test_normalised_tiles_share_one_stain_basis():
    a = om.synthetic_he_tiles(1, 128, seed=1)[0]
    b = om.synthetic_he_tiles(1, 128, seed=1, he=[[0.55, 0.15], [0.75, 0.90], [0.36, 0.20]])[0]       # the same tissue, a different stain batch
    na, nb = om.macenko_normalize(a), om.macenko_normalize(b)
    before = np.abs(a.astype(float) - b.astype(float)).mean()
    after = np.abs(na.astype(float) - nb.astype(float)).mean()
    assert after < 0.9 * before
    ha, _ = om.macenko_fit(na)
    assert _ang(ha[:, 0], om.HE_REF[:, 0]) < 6.0 and _ang(ha[:, 1], om.HE_REF[:, 1]) < 6.0
    white = np.full((32, 32, 3), 250, np.uint8)
    assert np.array_equal(om.macenko_normalize(white), white)                                     # nothing stained: passed through
