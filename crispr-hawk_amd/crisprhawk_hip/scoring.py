"""scoring — CFDon / DeepCpf1 / Azimuth slices of the reference's scoring.py with the arithmetic
on the GPU.  Model parameters are explicit inputs (``set_cfd_tables`` / ``load_cfd_tables``): the
reference downloads them from Zenodo at run time (config_utils.py:34-70) which is not possible
offline; loaders accept the reference's pickle format.  ``threads`` is ignored (a HIP context does
not survive the ProcessPoolExecutor fork of scoring.py:129)."""
import ctypes as C
import os
import pickle
from typing import Dict, List, Optional, Tuple, Union

import numpy as np

from . import _lib
from .crisprhawk_error import CrisprHawkAzimuthScoreError, CrisprHawkCfdScoreError, CrisprHawkDeepCpf1ScoreError
from .exception_handlers import exception_handler
from .guide import GUIDESEQPAD, Guide
from .pam import CPF1, PAM, SPCAS9, XCAS9
from .utils import VERBOSITYLVL, flatten_list, print_verbosity

_CFD_TABLES: Optional[Tuple[np.ndarray, np.ndarray]] = None
_RNA, _DNA = "ACGU", "ACGT"
_RCD = {"A": "T", "C": "G", "G": "C", "T": "A"}


def set_cfd_tables(mm: np.ndarray, pam: np.ndarray) -> None:
    """mm[20,4,4] = [position, wildtype RNA base, sgRNA base], pam[16] = PAM[-2:] dinucleotide."""
    global _CFD_TABLES
    _CFD_TABLES = (np.ascontiguousarray(mm, dtype=np.float64).reshape(20, 4, 4),
                   np.ascontiguousarray(pam, dtype=np.float64).reshape(16))


def cfd_tables_from_dicts(mmscores: Dict[str, float], pamscores: Dict[str, float]) -> Tuple[np.ndarray, np.ndarray]:
    """The reference's two dicts (keys ``r{wt}:d{RC(sg)},{i}`` and ``{pam}``, cfdscore.py:88-94) -> arrays.
    Entries the reference never looks up (wt == sg) may be absent; they are set to 1."""
    mm = np.ones((20, 4, 4))
    for i in range(20):
        for a in range(4):
            for b in range(4):
                k = f"r{_RNA[a]}:d{_RCD[_DNA[b]]},{i + 1}"
                if k in mmscores:
                    mm[i, a, b] = mmscores[k]
    pam = np.array([pamscores[_DNA[a] + _DNA[b]] for a in range(4) for b in range(4)], dtype=np.float64)
    return mm, pam


def load_cfd_tables(modelspath: str, debug: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """cfdscore.load_mismatch_pam_scores (cfdscore.py:22-50): mismatch_score.pkl + pam_scores.pkl.  The tables become
    the module's current ones and are returned as (mm[20,4,4], pam[16])."""
    try:
        with open(os.path.join(modelspath, "mismatch_score.pkl"), "rb") as f:
            mmscores = pickle.load(f)
        with open(os.path.join(modelspath, "pam_scores.pkl"), "rb") as f:
            pamscores = pickle.load(f)
    except OSError as e:
        exception_handler(CrisprHawkCfdScoreError, "An error occurred while loading CFD model files", os.EX_NOINPUT, debug, e)
    set_cfd_tables(*cfd_tables_from_dicts(mmscores, pamscores))
    return _CFD_TABLES


def _tables(debug: bool):
    if _CFD_TABLES is None:
        exception_handler(CrisprHawkCfdScoreError, "An error occurred while loading CFD model files", os.EX_NOINPUT, debug)
    return _CFD_TABLES


def _extract_guide_sequences(guides: List[Guide]) -> List[str]:
    """scoring.py:50-67: 4 nt upstream + guide/PAM + 3 nt downstream, upper case."""
    return [g.sequence[(GUIDESEQPAD - 4):(-GUIDESEQPAD + 3)].upper() for g in guides]


def group_guides_position(guides: List[Guide], debug: bool):
    """scoring.py:303-349"""
    groups: Dict[str, list] = {}
    for g in guides:
        key = f"{g.start}_{g.strand}"
        grp = groups.setdefault(key, [None, []])
        if g.samples == "REF":
            if grp[0] is not None:
                exception_handler(CrisprHawkCfdScoreError,
                                  f"Duplicate REF guide at position {g.start}? CFDon/Elevation-on calculation failed",
                                  os.EX_DATAERR, debug)
            grp[0] = g
        grp[1].append(g)
    return {k: (v[0], v[1]) for k, v in groups.items()}


def compute_cfd_batch(wt: List[str], sg: List[str], pam2: List[str], debug: bool) -> np.ndarray:
    """compute_cfd (cfdscore.py:53-95) for n (wildtype, sgRNA, PAM[-2:]) triples on the GPU.  Triples of one length go
    up as one batch; off-target rows with a bulge are one base longer than the rest and form a batch of their own."""
    mm, pt = _tables(debug)
    n = len(wt)
    out = np.empty(n, dtype=np.float64)
    if n == 0:
        return out
    if any(len(a) != len(b) for a, b in zip(wt, sg)) or any(len(p) != 2 for p in pam2):
        exception_handler(CrisprHawkCfdScoreError, "CFDon score calculation failed", os.EX_DATAERR, debug)
    lens = np.fromiter((len(x) for x in wt), dtype=np.int64, count=n)
    for ln in np.unique(lens):
        idx = np.flatnonzero(lens == ln)
        part = np.empty(len(idx), dtype=np.float64)
        whole = len(idx) == n
        w = wt if whole else [wt[i] for i in idx]
        g = sg if whole else [sg[i] for i in idx]
        p = pam2 if whole else [pam2[i] for i in idx]
        rc = _lib.lib().hawk_cfd(_lib.context(), "".join(w).encode("ascii"), "".join(g).encode("ascii"), int(ln),
                                 "".join(p).encode("ascii"), C.c_uint64(len(idx)), mm.ctypes.data_as(C.c_void_p),
                                 pt.ctypes.data_as(C.c_void_p), part.ctypes.data_as(C.c_void_p))
        if rc == _lib.HAWK_E_CFD:
            exception_handler(CrisprHawkCfdScoreError, "CFDon score calculation failed", os.EX_DATAERR, debug)
        _lib.check(rc, "hawk_cfd")
        out[idx] = part
    return out


def cfdon(guide_ref: Union[None, Guide], guides: List[Guide], debug: bool) -> List[float]:
    """scores/crisprhawk_scores.py:65-87"""
    if not guide_ref:
        return [np.nan] * len(guides)
    return compute_cfd_batch([guide_ref.guide] * len(guides), [g.guide for g in guides], [g.pam[-2:] for g in guides],
                             debug).tolist()


def cfdon_score(guides: List[Guide], verbosity: int, debug: bool) -> List[Guide]:
    """scoring.py:352-387: one device batch for all groups; returns the guides in group order."""
    print_verbosity("Computing CFDon score", verbosity, VERBOSITYLVL[3])
    groups = group_guides_position(guides, debug)
    wt, sg, pm, dst = [], [], [], []
    for _, (ref, members) in groups.items():
        for g in members:
            if ref is None:
                g.cfdon_score = float("nan")
            else:
                wt.append(ref.guide); sg.append(g.guide); pm.append(g.pam[-2:]); dst.append(g)
    if dst:
        for g, s in zip(dst, compute_cfd_batch(wt, sg, pm, debug).tolist()):
            g.cfdon_score = float(s)
    return flatten_list([members for _, (_, members) in groups.items()])


# ---------------------------------------------------------------------------- DeepCpf1 (K6)
_DEEPCPF1_W: Optional[np.ndarray] = None
_DC_KEYS = ("conv_w", "conv_b", "w1", "b1", "w2", "b2", "w3", "b3", "w4", "b4")
_DC_SHAPES = ((80, 4, 5), (80,), (80, 1200), (80,), (40, 80), (40,), (40, 40), (40,), (1, 40), (1,))


def set_deepcpf1_weights(w: Dict[str, np.ndarray]) -> None:
    """Parameters in the torch layout of the reference's SeqDeepCpf1 (seqdeepcpf1.py:43-56):
    conv_w (80,4,5), conv_b, w1 (80,1200), b1, w2 (40,80), b2, w3 (40,40), b3, w4 (1,40), b4."""
    global _DEEPCPF1_W
    parts = []
    for k, shp in zip(_DC_KEYS, _DC_SHAPES):
        a = np.ascontiguousarray(w[k], dtype=np.float32)
        if a.shape != shp:
            raise ValueError(f"DeepCpf1 parameter {k} has shape {a.shape}, expected {shp}")
        parts.append(a.reshape(-1))
    _DEEPCPF1_W = np.concatenate(parts)


def deepcpf1_weights_from_keras(kw: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """The Keras arrays of Seq_deepCpf1_weights.h5 -> torch layout, with the transposes / kernel
    flip of load_deepcpf1_weights (seqdeepcpf1.py:113-124).  Keys: convolution1d_157_W/_b,
    dense_490.._493 _W/_b (read them with any HDF5 reader; h5py is not a dependency here)."""
    cw = np.asarray(kw["convolution1d_157_W"], dtype=np.float32)  # (5, 1, 4, 80)
    cw = np.flip(np.transpose(np.squeeze(cw, 1), (2, 1, 0)), -1)
    out = {"conv_w": cw, "conv_b": kw["convolution1d_157_b"]}
    for i, name in enumerate(("dense_490", "dense_491", "dense_492", "dense_493")):
        out[f"w{i + 1}"] = np.asarray(kw[f"{name}_W"], dtype=np.float32).T
        out[f"b{i + 1}"] = kw[f"{name}_b"]
    return out


def deepcpf1(guides: List[str], debug: bool = True) -> List[float]:
    """scores/crisprhawk_scores.py:90-107 on the GPU: 34-mers -> scores."""
    if _DEEPCPF1_W is None:
        exception_handler(CrisprHawkDeepCpf1ScoreError, "DeepCpf1 weights not loaded (scoring.set_deepcpf1_weights)",
                          os.EX_NOINPUT, debug)
    n = len(guides)
    if n == 0:
        return []
    if any(len(gd) != 34 for gd in guides):
        exception_handler(CrisprHawkDeepCpf1ScoreError, "DeepCpf1 score calculation failed (34-nt inputs required)",
                          os.EX_DATAERR, debug)
    out = np.empty(n, dtype=np.float32)
    rc = _lib.lib().hawk_deepcpf1(_lib.context(), "".join(guides).encode("ascii"), C.c_uint64(n),
                                  _DEEPCPF1_W.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    if rc == _lib.HAWK_E_IUPAC:
        exception_handler(CrisprHawkDeepCpf1ScoreError, "DeepCpf1 score calculation failed", os.EX_DATAERR, debug)
    _lib.check(rc, "hawk_deepcpf1")
    return [float(x) for x in out]


def deepcpf1_score(guides: List[Guide], threads: int, verbosity: int, debug: bool) -> List[Guide]:
    """scoring.py:456-497 (one GPU batch; ``threads`` is ignored)."""
    if not guides:
        return guides
    print_verbosity("Computing DeepCpf1 score", verbosity, VERBOSITYLVL[3])
    for g, s in zip(guides, deepcpf1(_extract_guide_sequences(guides), debug)):
        g.deepcpf1_score = s
    return guides


# ---------------------------------------------------------------------------- Azimuth (K5)
_AZIMUTH_MODEL: Optional[Dict] = None


def azimuth_model_from_sklearn(gbr) -> Dict:
    """Flatten a fitted sklearn GradientBoostingRegressor (what the reference unpickles from
    saved_models/V3_model_nopos.pickle, model_comparison.py:538-550) into the arrays hawk_azimuth takes."""
    tree_off, feature, left, right, thr, val = [0], [], [], [], [], []
    for est in gbr.estimators_[:, 0]:
        t = est.tree_
        for k in range(t.node_count):
            leaf = t.children_left[k] == -1
            feature.append(-1 if leaf else int(t.feature[k]))
            left.append(0 if leaf else int(t.children_left[k]))
            right.append(0 if leaf else int(t.children_right[k]))
            thr.append(float(t.threshold[k]))
            val.append(float(t.value[k, 0, 0]))
        tree_off.append(len(feature))
    init = gbr.init_
    init_val = float(init.constant_[0, 0]) if hasattr(init, "constant_") else float(np.ravel(init.predict(np.zeros((1, gbr.n_features_in_))))[0])
    return dict(tree_off=np.array(tree_off, np.int32), feature=np.array(feature, np.int32), left=np.array(left, np.int32),
                right=np.array(right, np.int32), threshold=np.array(thr, np.float64), value=np.array(val, np.float64),
                init=init_val, learning_rate=float(gbr.learning_rate))


def set_azimuth_model(model) -> None:
    """A dict of flattened arrays (see azimuth_model_from_sklearn) or a fitted sklearn GBR."""
    global _AZIMUTH_MODEL
    _AZIMUTH_MODEL = model if isinstance(model, dict) else azimuth_model_from_sklearn(model)


def tm_nn(seqs) -> List[float]:
    """Bio.SeqUtils.MeltingTemp.Tm_NN with Biopython's defaults for equally long A/C/G/T strings of <= 32 bases, on the device
    (hawk_tm_nn: the function behind Azimuth's four Tm features, featurization.py:358-397)."""
    seqs = [str(x) for x in seqs]
    if not seqs:
        return []
    ln = len(seqs[0])
    if any(len(x) != ln for x in seqs):
        raise ValueError("tm_nn: sequences of one length per call")
    out = np.empty(len(seqs), dtype=np.float64)
    _lib.check(_lib.lib().hawk_tm_nn(_lib.context(), "".join(seqs).encode("ascii"), C.c_uint32(ln), C.c_uint64(len(seqs)),
                                     out.ctypes.data_as(C.c_void_p)), "hawk_tm_nn")
    return list(out)


def azimuth(guides, debug: bool = True, return_features: bool = False):
    """scores/crisprhawk_scores.py:31-44 (azimuth.model_comparison.predict) on the GPU: 30-mers -> scores."""
    if _AZIMUTH_MODEL is None:
        exception_handler(CrisprHawkAzimuthScoreError, "Azimuth model not loaded (scoring.set_azimuth_model)", os.EX_NOINPUT, debug)
    guides = [str(x) for x in guides]
    n = len(guides)
    if n == 0:
        return []
    if any(len(gd) != 30 for gd in guides):
        exception_handler(CrisprHawkAzimuthScoreError, "Azimuth score calculation failed (30-nt inputs required)", os.EX_DATAERR, debug)
    m = _AZIMUTH_MODEL
    arrs = {k: np.ascontiguousarray(m[k], dtype=(np.float64 if k in ("threshold", "value") else np.int32))
            for k in ("tree_off", "feature", "left", "right", "threshold", "value")}
    gm = _lib.GbtModel(len(arrs["tree_off"]) - 1, len(arrs["feature"]), *[arrs[k].ctypes.data for k in
                       ("tree_off", "feature", "left", "right", "threshold", "value")], float(m["init"]), float(m["learning_rate"]))
    out = np.empty(n, dtype=np.float64)
    feats = np.empty((n, 627), dtype=np.float64) if return_features else None
    rc = _lib.lib().hawk_azimuth(_lib.context(), "".join(guides).encode("ascii"), C.c_uint64(n), C.byref(gm),
                                 out.ctypes.data_as(C.c_void_p), feats.ctypes.data_as(C.c_void_p) if return_features else None)
    if rc == _lib.HAWK_E_IUPAC:
        exception_handler(CrisprHawkAzimuthScoreError, "Azimuth score calculation failed", os.EX_DATAERR, debug)
    _lib.check(rc, "hawk_azimuth")
    return (list(out), feats) if return_features else list(out)


def azimuth_score(guides: List[Guide], threads: int, verbosity: int, debug: bool) -> List[Guide]:
    """scoring.py:152-193 (one GPU batch; ``threads`` is ignored)."""
    if not guides:
        return guides
    print_verbosity("Computing Azimuth score", verbosity, VERBOSITYLVL[3])
    for g, s in zip(guides, azimuth(_extract_guide_sequences(guides), debug)):
        g.azimuth_score = float(s)
    return guides


# ---------------------------------------------------------------------------------------------- RS3 (a20)
_RS3: Optional[Tuple[Dict, object]] = None  # (flattened LightGBM model, featuriser)


def gbt_model_from_lightgbm_text(text: str) -> Dict:
    """A LightGBM model in its text format (Booster.save_model / model_to_string) -> the flattened tree arrays of
    hawk_gbt_model.  Internal node i of a tree keeps index i; leaf j becomes node num_internal + j (LightGBM writes a
    child c < 0 for leaf ~c).  Numerical splits only, default direction and missing values are not modelled (RS3's
    features are dense one-hots, counts and floats); `shrinkage` is already folded into the leaf values."""
    trees = []
    cur: Dict[str, str] = {}
    for line in text.splitlines():
        line = line.strip()
        if line.startswith("Tree="):
            cur = {}
            trees.append(cur)
        elif line == "end of trees":
            break
        elif "=" in line and trees:
            k, v = line.split("=", 1)
            cur[k] = v
    if not trees:
        raise ValueError("no Tree= sections: not a LightGBM text model")
    tree_off, feature, left, right, threshold, value = [0], [], [], [], [], []
    n_feat = 0
    for t in trees:
        leaves = [float(x) for x in t["leaf_value"].split()]
        n_int = int(t["num_leaves"]) - 1
        if n_int == 0:
            feature += [-1]; left += [0]; right += [0]; threshold += [0.0]; value += [leaves[0]]
            tree_off.append(len(feature))
            continue
        if any(int(d) & 1 for d in t.get("decision_type", "").split()):
            raise ValueError("categorical splits are not supported")
        sf = [int(x) for x in t["split_feature"].split()]
        th = [float(x) for x in t["threshold"].split()]
        lc = [int(x) for x in t["left_child"].split()]
        rc = [int(x) for x in t["right_child"].split()]
        n_feat = max(n_feat, max(sf) + 1)
        # the device walker wants children AFTER their parent: renumber in breadth-first order
        order, seen = [0], {0}
        for node in order:
            if node < n_int:
                for c in (lc[node], rc[node]):
                    c = c if c >= 0 else n_int + (~c)
                    if c not in seen:
                        seen.add(c)
                        order.append(c)
        new = {old: i for i, old in enumerate(order)}
        rows = [None] * len(order)
        for old, i in new.items():
            if old < n_int:
                l_, r_ = (c if c >= 0 else n_int + (~c) for c in (lc[old], rc[old]))
                rows[i] = (sf[old], new[l_], new[r_], th[old], 0.0)
            else:
                rows[i] = (-1, 0, 0, 0.0, leaves[old - n_int])
        for f_, l_, r_, th_, v_ in rows:
            feature.append(f_); left.append(l_); right.append(r_); threshold.append(th_); value.append(v_)
        tree_off.append(len(feature))
    return dict(tree_off=np.array(tree_off, np.int32), feature=np.array(feature, np.int32), left=np.array(left, np.int32),
                right=np.array(right, np.int32), threshold=np.array(threshold, np.float64), value=np.array(value, np.float64),
                init=0.0, learning_rate=1.0, n_features=n_feat)


def set_rs3_model(model, featurizer) -> None:
    """`model`: a LightGBM text model (str), or a flattened dict; `featurizer(kmers) -> [n, n_features] array` is the
    caller's sglearn feature builder (rs3.seq.predict_seq featurises with `sglearn.featurize_guides`, third party)."""
    global _RS3
    m = gbt_model_from_lightgbm_text(model) if isinstance(model, str) else dict(model)
    _RS3 = (m, featurizer)


def gbt_predict(feats: np.ndarray, model: Dict, cast_f32: bool = False) -> np.ndarray:
    """hawk_gbt_predict: a flattened tree ensemble over a feature matrix, on the device."""
    x = np.ascontiguousarray(feats, dtype=np.float64)
    n, nf = x.shape
    arrs = {k: np.ascontiguousarray(model[k], dtype=(np.float64 if k in ("threshold", "value") else np.int32))
            for k in ("tree_off", "feature", "left", "right", "threshold", "value")}
    gm = _lib.GbtModel(len(arrs["tree_off"]) - 1, len(arrs["feature"]), *[arrs[k].ctypes.data for k in
                       ("tree_off", "feature", "left", "right", "threshold", "value")], float(model["init"]), float(model["learning_rate"]))
    out = np.empty(n, dtype=np.float64)
    _lib.check(_lib.lib().hawk_gbt_predict(_lib.context(), x.ctypes.data_as(C.c_void_p), C.c_uint64(n), nf, C.byref(gm), int(cast_f32),
                                           out.ctypes.data_as(C.c_void_p)), "hawk_gbt_predict")
    return out


def rs3(guides: List[str], debug: bool = True) -> List[float]:
    """scores/crisprhawk_scores.py:47-62 (rs3.seq.predict_seq, tracr Hsu2013): 30-mers -> scores.  The feature matrix
    comes from the caller's featuriser, the LightGBM trees run on the device."""
    from .crisprhawk_error import CrisprHawkRs3ScoreError
    if _RS3 is None:
        exception_handler(CrisprHawkRs3ScoreError, "RS3 model not loaded (scoring.set_rs3_model)", os.EX_NOINPUT, debug)
    model, featurizer = _RS3
    if not guides:
        return []
    try:
        feats = np.asarray(featurizer(list(guides)), dtype=np.float64)
    except Exception as e:
        exception_handler(CrisprHawkRs3ScoreError, "RS3 score calculation failed", os.EX_DATAERR, debug, e)
    if feats.ndim != 2 or feats.shape[0] != len(guides) or feats.shape[1] < int(model.get("n_features", 0)):
        exception_handler(CrisprHawkRs3ScoreError, "RS3 featuriser returned a matrix of the wrong shape", os.EX_DATAERR, debug)
    return list(gbt_predict(feats, model, cast_f32=False))


def rs3_score(guides: List[Guide], threads: int, verbosity: int, debug: bool) -> List[Guide]:
    """scoring.py:261-300 (one device batch; ``threads`` is ignored).  Without a model rs3() raises CrisprHawkRs3ScoreError."""
    if not guides:
        return guides
    print_verbosity("Computing RS3 score", verbosity, VERBOSITYLVL[3])
    for g, s in zip(guides, rs3(_extract_guide_sequences(guides), debug)):
        g.rs3_score = float(s)
    return guides


# ---------------------------------------------------------------------------------------------- model files (f4)
def load_models(models_dir: str, debug: bool = True) -> Dict[str, bool]:
    """Load whatever scorer parameters `models_dir` holds and make them current.  Accepted files:

        mismatch_score.pkl + pam_scores.pkl   the reference's CFD pickles (plain dicts; cfdscore.py:22-50)
        cfd_tables.npz                        mm[20,4,4], pam[16]
        azimuth_model.npz                     tree_off, feature, left, right, threshold, value, init, learning_rate
        deepcpf1_weights.npz                  conv_w, conv_b, w1, b1, w2, b2, w3, b3, w4, b4 (torch layout)
        rs3_model.txt                         LightGBM text model (needs scoring.set_rs3_model's featuriser separately)

    The .npz files are what tools/convert_models.py writes from the reference's own downloads (the sklearn pickle and the
    Keras .h5 need scikit-learn / h5py, which only the machine that fetched the models is sure to have).
    Returns which scorers are now available."""
    have = {"cfd": False, "azimuth": False, "deepcpf1": False, "rs3_model": False}
    j = lambda f: os.path.join(models_dir, f)
    if os.path.exists(j("cfd_tables.npz")):
        z = np.load(j("cfd_tables.npz"))
        set_cfd_tables(z["mm"], z["pam"])
        have["cfd"] = True
    elif os.path.exists(j("mismatch_score.pkl")) and os.path.exists(j("pam_scores.pkl")):
        load_cfd_tables(models_dir, debug)
        have["cfd"] = True
    if os.path.exists(j("azimuth_model.npz")):
        z = np.load(j("azimuth_model.npz"))
        set_azimuth_model({k: (float(z[k]) if k in ("init", "learning_rate") else z[k]) for k in z.files})
        have["azimuth"] = True
    if os.path.exists(j("deepcpf1_weights.npz")):
        z = np.load(j("deepcpf1_weights.npz"))
        set_deepcpf1_weights({k: z[k] for k in _DC_KEYS})
        have["deepcpf1"] = True
    if os.path.exists(j("rs3_model.txt")):
        have["rs3_model"] = True
    return have


# Scorers the reference calls that have no counterpart here (external checkpoints / conda environments).
# Their columns keep "NA"; they are listed so that scoring_guides documents, in code, what it leaves out.
OUT_OF_SCOPE_SCORERS = ("plmcrispr", "crispron", "sgdesigner")
_SKIP: set = set()


def skip_scorers(*names: str) -> None:
    """Opt out of in-scope scorers by name ("azimuth", "rs3", "cfdon", "deepcpf1") for callers that do not hold their
    parameters.  Nothing is skipped by default: like the reference, scoring_guides calls every scorer of the Cas system
    and a scorer without its model raises its CrisprHawk*ScoreError.  skip_scorers() with no argument clears the set."""
    bad = set(names) - {"azimuth", "rs3", "cfdon", "deepcpf1"}
    if bad:
        raise ValueError(f"unknown scorer(s): {sorted(bad)}")
    _SKIP.clear()
    _SKIP.update(names)


def _scoring_guides_cas9(guides_list: List[Guide], cas_system: int, scoring_envs, threads: int, verbosity: int, debug: bool) -> List[Guide]:
    """scoring.py:749-792: azimuth, rs3, (plm-crispr), cfdon, (crispron, sgdesigner) - in that order; cfdon_score
    returns the guides in group order (scoring.py:383), so the list order changes there exactly as in the reference."""
    if "azimuth" not in _SKIP:
        guides_list = azimuth_score(guides_list, threads, verbosity, debug)
    if "rs3" not in _SKIP:
        guides_list = rs3_score(guides_list, threads, verbosity, debug)
    if "cfdon" not in _SKIP:
        guides_list = cfdon_score(guides_list, verbosity, debug)
    return guides_list


def _scoring_guides_cpf1(guides_list: List[Guide], threads: int, verbosity: int, debug: bool) -> List[Guide]:
    """scoring.py:795-813"""
    if "deepcpf1" in _SKIP:
        return guides_list
    return deepcpf1_score(guides_list, threads, verbosity, debug)


def scoring_guides(guides: Dict, pam: PAM, scoring_envs, args) -> Dict:
    """scoring.py:816-867: SpCas9 / xCas9 PAMs -> azimuth, rs3, cfdon; Cpf1 PAMs -> deepcpf1; any other PAM -> no score.
    A scorer whose parameters were never supplied raises its own error class, as the reference's does when its model
    file is missing; Elevation-on (an external package with its own checkpoints) is refused rather than left at NA."""
    print_verbosity("Scoring guides", args.verbosity, VERBOSITYLVL[1])
    for region, guides_list in guides.items():
        if pam.cas_system in (SPCAS9, XCAS9):
            guides_list = _scoring_guides_cas9(guides_list, pam.cas_system, scoring_envs, args.threads, args.verbosity, args.debug)
        elif pam.cas_system == CPF1:
            guides_list = _scoring_guides_cpf1(guides_list, args.threads, args.verbosity, args.debug)
        if getattr(args, "compute_elevation", False) and (args.guidelen + len(pam) == 23 and not args.right):
            from .crisprhawk_error import CrisprHawkElevationScoreError
            exception_handler(CrisprHawkElevationScoreError, "Elevation-on is not part of the GPU scoring path", os.EX_DATAERR, args.debug)
        guides[region] = guides_list
    return guides
