"""model-file handling.  CPU: the LightGBM text-model flattener (RS3's boundary) against a direct walk
of the same text's trees, and scoring.load_models on the file formats tools/convert_models.py writes.  GPU: the device tree
evaluator over a supplied feature matrix, and - only where HAWK_MODELS_DIR points at converted real models - the
reference's own known answers (/root/reference/tests/test_scoring.py:55-76: Azimuth 0.472, RS3 -0.996, DeepCpf1 47.265)."""
import os

import numpy as np
import pytest

from crisprhawk_hip import scoring, synth

LGB_TEXT = """tree
version=v3
num_class=1
num_tree_per_iteration=1
label_index=0
max_feature_idx=5
objective=regression
feature_names=f0 f1 f2 f3 f4 f5
feature_infos=none none none none none none
tree_sizes=300 200

Tree=0
num_leaves=4
num_cat=0
split_feature=2 0 5
split_gain=1 1 1
threshold=0.5 1.25 -0.75
decision_type=2 2 2
left_child=1 -1 -3
right_child=2 -2 -4
leaf_value=0.1 -0.2 0.3 0.45
leaf_weight=1 1 1 1
leaf_count=1 1 1 1
internal_value=0 0 0
internal_weight=0 0 0
internal_count=4 2 2
shrinkage=1

Tree=1
num_leaves=3
num_cat=0
split_feature=4 1
split_gain=1 1
threshold=0.0 2.0
decision_type=2 2
left_child=-1 -2
right_child=1 -3
leaf_value=-0.05 0.02 0.07
shrinkage=0.1

Tree=2
num_leaves=1
num_cat=0
leaf_value=0.5
shrinkage=0.1

end of trees

feature_importances:
f2=1
"""


def _walk_text(x):
    """LightGBM's own traversal of LGB_TEXT: child < 0 is leaf ~child; go left when x[feature] <= threshold."""
    t0 = dict(sf=[2, 0, 5], th=[0.5, 1.25, -0.75], lc=[1, -1, -3], rc=[2, -2, -4], lv=[0.1, -0.2, 0.3, 0.45])
    t1 = dict(sf=[4, 1], th=[0.0, 2.0], lc=[-1, -2], rc=[1, -3], lv=[-0.05, 0.02, 0.07])
    tot = 0.5
    for t in (t0, t1):
        node = 0
        while node >= 0:
            node = t["lc"][node] if x[t["sf"][node]] <= t["th"][node] else t["rc"][node]
        tot += t["lv"][~node]
    return tot


def _cpu_eval(model, X):
    out = np.full(len(X), model["init"])
    for i, x in enumerate(X):
        for t in range(len(model["tree_off"]) - 1):
            base = node = int(model["tree_off"][t])
            while model["feature"][node] >= 0:
                node = base + int(model["left"][node] if x[model["feature"][node]] <= model["threshold"][node] else model["right"][node])
            out[i] += model["learning_rate"] * model["value"][node]
    return out


def test_lightgbm_text_model_flattening():
    m = scoring.gbt_model_from_lightgbm_text(LGB_TEXT)
    assert len(m["tree_off"]) == 4 and m["n_features"] == 6
    for t in range(3):  # children come after their parents: the device walker's termination argument
        lo, hi = int(m["tree_off"][t]), int(m["tree_off"][t + 1])
        for k in range(lo, hi):
            if m["feature"][k] >= 0:
                assert k - lo < m["left"][k] < hi - lo and k - lo < m["right"][k] < hi - lo
    rng = np.random.default_rng(3)
    X = rng.normal(0, 1.5, size=(500, 6))
    X[:50, 2] = 0.5  # ties go left
    want = np.array([_walk_text(x) for x in X])
    assert np.allclose(_cpu_eval(m, X), want, rtol=0, atol=1e-15)
    with pytest.raises(ValueError):
        scoring.gbt_model_from_lightgbm_text("not a model")


def test_load_models_reads_converted_files(tmp_path):
    mm, pt = synth.cfd_tables()
    np.savez(tmp_path / "cfd_tables.npz", mm=mm, pam=pt)
    w = synth.deepcpf1_weights()
    np.savez(tmp_path / "deepcpf1_weights.npz", **w)
    m = scoring.gbt_model_from_lightgbm_text(LGB_TEXT)
    np.savez(tmp_path / "azimuth_model.npz", **{k: m[k] for k in ("tree_off", "feature", "left", "right", "threshold", "value", "init", "learning_rate")})
    (tmp_path / "rs3_model.txt").write_text(LGB_TEXT)
    have = scoring.load_models(str(tmp_path))
    assert have == {"cfd": True, "azimuth": True, "deepcpf1": True, "rs3_model": True}
    assert np.array_equal(scoring._CFD_TABLES[0], mm) and scoring._AZIMUTH_MODEL is not None and scoring._DEEPCPF1_W is not None
    # the reference's pickle pair is accepted as well (cfdscore.py:22-50)
    import pickle
    mmd, pamd = synth.cfd_tables_as_dicts(mm, pt)
    d2 = tmp_path / "pk"
    d2.mkdir()
    pickle.dump(mmd, open(d2 / "mismatch_score.pkl", "wb"))
    pickle.dump(pamd, open(d2 / "pam_scores.pkl", "wb"))
    assert scoring.load_models(str(d2))["cfd"] and np.allclose(scoring._CFD_TABLES[1], pt)


def test_rs3_boundary_without_model_raises():
    """The reference's rs3_score ends in CrisprHawkRs3ScoreError when the scorer cannot run (scoring.py:261-300)."""
    from crisprhawk_hip.crisprhawk_error import CrisprHawkRs3ScoreError
    from crisprhawk_hip.guide import Guide
    scoring._RS3 = None
    g = Guide(1, 24, "C" * 10 + "AGCTTAGCTAGCTAGCTAGCTAG" + "C" * 10, 20, 3, 0, "REF", "NA", {}, {i: i for i in range(43)}, True, False, "hap1")
    with pytest.raises(CrisprHawkRs3ScoreError):
        scoring.rs3_score([g], 1, 0, True)
    assert g.rs3_score == "NA" and scoring.rs3_score([], 1, 0, True) == []
    with pytest.raises(CrisprHawkRs3ScoreError):
        scoring.rs3(["A" * 30], True)


@pytest.mark.gpu
def test_device_gbt_over_supplied_features():
    m = scoring.gbt_model_from_lightgbm_text(LGB_TEXT)
    rng = np.random.default_rng(4)
    X = rng.normal(0, 1.5, size=(4000, 6))
    assert np.array_equal(scoring.gbt_predict(X, m), _cpu_eval(m, X))
    scoring.set_rs3_model(LGB_TEXT, lambda kmers: np.array([[ord(c) % 5 - 2.0 for c in k[:6]] for k in kmers]))
    kmers = ["ACGTAC" + "A" * 24, "TTTTTT" + "C" * 24]
    feats = np.array([[ord(c) % 5 - 2.0 for c in k[:6]] for k in kmers])
    assert np.array_equal(np.array(scoring.rs3(kmers)), _cpu_eval(m, feats))
    scoring._RS3 = None


_MODELS = os.environ.get("HAWK_MODELS_DIR")


@pytest.mark.gpu
@pytest.mark.skipif(not _MODELS, reason="HAWK_MODELS_DIR not set: the reference's scoring models (Zenodo downloads) are not available offline")
def test_reference_known_answers_with_real_models():
    have = scoring.load_models(_MODELS)
    k30 = ("C" * 10 + "AGCTTAGCTAGCTAGCTAGCTAG" + "C" * 10)[6:-7]
    k34 = ("C" * 10 + "AGCTTAGCTAGCTAGCTAGCTAGTTTC" + "C" * 10)[6:-7]
    if have["azimuth"]:
        assert round(float(scoring.azimuth([k30])[0]), 3) == 0.472
    if have["deepcpf1"]:
        assert round(float(scoring.deepcpf1([k34])[0]), 3) == 47.265
    if have["rs3_model"]:
        sglearn = pytest.importorskip("sglearn")
        import pandas as pd
        text = open(os.path.join(_MODELS, "rs3_model.txt")).read()
        scoring.set_rs3_model(text, lambda ks: sglearn.featurize_guides(ks).values if hasattr(sglearn, "featurize_guides") else None)
        assert round(float(scoring.rs3([k30])[0]), 3) == -0.996
    assert any(have.values())
