#!/usr/bin/env python3
"""One-off converter, run by the user on the machine that holds the reference's downloaded scoring models
(`crisprhawk` fetches them from Zenodo at run time, config_utils.py:34-70; that machine has scikit-learn / h5py /
lightgbm because the reference needs them).  Writes dependency-free files crisprhawk_hip.scoring.load_models() reads
with numpy alone:

    cfd_tables.npz         mm[20,4,4] (position, wildtype RNA base A,C,G,U, sgRNA base A,C,G,T), pam[16] (PAM[-2:], 4*b0+b1)
                           <- scores/cfdscore/models/mismatch_score.pkl + pam_scores.pkl
    azimuth_model.npz      tree_off[T+1], feature[N] (-1 = leaf), left[N], right[N] (node indices relative to their
                           tree), threshold[N], value[N], init, learning_rate
                           <- azimuth/saved_models/V3_model_nopos.pickle (sklearn GradientBoostingRegressor)
    deepcpf1_weights.npz   conv_w[80,4,5] conv_b[80] w1[80,1200] b1[80] w2[40,80] b2[40] w3[40,40] b3[40] w4[1,40] b4[1]
                           (torch layout of SeqDeepCpf1) <- deepCpf1/weights/Seq_deepCpf1_weights.h5 (Keras)
    rs3_model.txt          LightGBM text model <- rs3's pickled booster (rs3 package data, RuleSet3.pkl)

    python tools/convert_models.py --models-dir <reference>/src/crisprhawk/scores --out <dir>
"""
import argparse
import glob
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crispr-hawk_amd")]

import numpy as np  # noqa: E402


def find(root, pattern):
    hits = glob.glob(os.path.join(root, "**", pattern), recursive=True)
    return hits[0] if hits else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models-dir", required=True)
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    from crisprhawk_hip import scoring
    os.makedirs(args.out, exist_ok=True)
    mmf, pamf = find(args.models_dir, "mismatch_score.pkl"), find(args.models_dir, "pam_scores.pkl")
    if mmf and pamf:
        mm, pam = scoring.cfd_tables_from_dicts(pickle.load(open(mmf, "rb")), pickle.load(open(pamf, "rb")))
        np.savez(os.path.join(args.out, "cfd_tables.npz"), mm=mm, pam=pam)
        print("cfd_tables.npz")
    azf = find(args.models_dir, "V3_model_nopos.pickle")
    if azf:
        obj = pickle.load(open(azf, "rb"))  # (model, learn_options), model_comparison.py:538-550
        gbr = obj[0] if isinstance(obj, (tuple, list)) else obj
        np.savez(os.path.join(args.out, "azimuth_model.npz"), **scoring.azimuth_model_from_sklearn(gbr))
        print("azimuth_model.npz")
    h5f = find(args.models_dir, "Seq_deepCpf1_weights.h5")
    if h5f:
        import h5py
        kw = {}
        with h5py.File(h5f, "r") as f:  # seqdeepcpf1.py:95-124 reads the same datasets
            f.visititems(lambda name, ds: kw.__setitem__(name.split("/")[-1], np.array(ds)) if hasattr(ds, "shape") else None)
        np.savez(os.path.join(args.out, "deepcpf1_weights.npz"), **scoring.deepcpf1_weights_from_keras(kw))
        print("deepcpf1_weights.npz")
    rsf = find(args.models_dir, "RuleSet3.pkl")
    if rsf:
        booster = pickle.load(open(rsf, "rb"))
        booster = getattr(booster, "booster_", booster)
        open(os.path.join(args.out, "rs3_model.txt"), "w").write(booster.model_to_string())
        print("rs3_model.txt")


if __name__ == "__main__":
    main()
