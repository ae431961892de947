"""Evaluation utilities of the reference on the device: landmark transfer through a displacement field and the landmark
accuracy measure -- same names, arguments and results as NNE/landmark_transform.m and NNE/landmark_accuracy.m
(cp_landmark_transform / cp_landmark_accuracy of include/cpengine.h) -- and the "evaluate results" / "save results"
sections of the demos (NNE/demo_mf_nn_curvature_registration_no_ref_ml.m:89-125)."""
import os
import warnings

import numpy as np

from . import engine as E


def landmark_transform(LM, u, omega, device=0):
    """LM_transformed = landmark_transform(LM, u, omega)   (NNE/landmark_transform.m:1-67)

    LM: (numLM, 2) landmark positions; u: (m, n, 2) displacement field. Solves x + u(x) = y by the fixed-point iteration
    x <- x + (y - F(x)) from the nearest deformed grid point, with the reference's literal termination expression
    `sum(F(x) - LM(i,:).^2) < 1e-12` (:50,:54) and its warning + fallback to the initial guess (:55-57)."""
    out, failed = E.landmark_transform(LM, u, omega, device)
    for _ in range(int(np.count_nonzero(failed))):
        warnings.warn("Fixed-point iteration failed! Returning initial guess.")
    return out


def landmark_accuracy(LM, device=0):
    """LM_acc = landmark_accuracy(LM)   (NNE/landmark_accuracy.m:1-28): LM is a list of k (numLM, 2) arrays; returns
    the mean distance of every landmark to its mean position over the k images."""
    return E.landmark_accuracy(LM, device)


def evaluate_results(img, uStar, L, omega=None, LM=None, device=0):
    """the "evaluate results" section of the demos (NNE/demo_mf_nn_curvature_registration_no_ref_ml.m:89-118):
    LStar = reshape(L, m, n, k); img_u{i} = evaluate_displacement(img{i}, [1, 1], uStar(:, :, i)); if landmarks exist,
    LM_transformed{i} = landmark_transform(LM{i}, reshape(uStar(:, :, i), [m n 2]), omega) and both accuracies.
    img: list of k (m, n) arrays; uStar: (m*n, 2, k); L: (m*n, k). Returns a dict with the demo's variable names."""
    k = len(img)
    m, n = np.asarray(img[0]).shape
    if omega is None:
        omega = [0.0, float(m), 0.0, float(n)]
    uStar = np.asarray(uStar, dtype=np.float64).reshape((m * n, 2, k), order="F")
    out = {"LStar": np.asarray(L, dtype=np.float64).reshape((m, n, k), order="F"),
           "img_u": [E.evaluate_displacement(img[i], [1.0, 1.0], uStar[:, :, i], device=device, want_grad=False)[0]
                     for i in range(k)]}
    if LM is not None:
        out["LM_transformed"] = [landmark_transform(LM[i], uStar[:, :, i].reshape((m, n, 2), order="F"), omega, device)
                                 for i in range(k)]
        out["LM_acc"] = landmark_accuracy(LM, device)
        out["LM_transformed_acc"] = landmark_accuracy(out["LM_transformed"], device)
    return out


def _cell(items):
    c = np.empty((len(items), 1), dtype=object)
    for i, a in enumerate(items):
        c[i, 0] = np.asarray(a, dtype=np.float64)
    return c


def save_results(directory, dataset, method, img, optPara, uStar, L, SV_history, runtime_registration, evaluated,
                 omega=None, LM=None):
    """the "save results" section of the demos (:120-125): `save(sprintf('./results/res_%s_<model>', dataset))` writes the
    demo's workspace to a MATLAB file. Written here as a version-5 MAT-file with the same variable names and shapes
    (cell arrays for img, img_u, LM, LM_transformed and SV_history; optPara as a struct), so that the reference's
    plotting sections (:127-173) and MATLAB's `load` read it unchanged. Returns the path.
    method: the demo's `mfilename`, e.g. 'demo_mf_nn_curvature_registration_no_ref_ml' -- the file name takes the model
    part ('mf_nn_curvature', 'mf_nn_tv', 'var_tv', 'var_curvature') as the demos do."""
    from scipy.io import savemat
    k = len(img)
    m, n = np.asarray(img[0]).shape
    model = method[len("demo_"):] if method.startswith("demo_") else method
    for tail in ("_registration_no_ref_ml", "_registration_no_ref"):
        if model.endswith(tail):
            model = model[: -len(tail)]
    os.makedirs(directory, exist_ok=True)                                   # :122-124
    path = os.path.join(directory, f"res_{dataset}_{model}.mat")
    ws = {"dataset": dataset, "method": method, "k": float(k), "m": float(m), "n": float(n),
          "omega": np.asarray(omega if omega is not None else [0.0, m, 0.0, n], dtype=np.float64).reshape(1, 4),
          "img": _cell(img), "optPara": {kk: (np.asarray(v, dtype=np.float64) if not isinstance(v, (str, bool)) else v)
                                         for kk, v in optPara.items()},
          "uStar": np.asarray(uStar, dtype=np.float64).reshape((m * n, 2, k), order="F"),
          "L": np.asarray(L, dtype=np.float64).reshape((m * n, k), order="F"),
          "SV_history": _cell(SV_history), "runtime_registration": float(runtime_registration),
          "LStar": evaluated["LStar"], "img_u": _cell(evaluated["img_u"]), "LM_available": LM is not None}
    if LM is not None:
        lm = np.empty((1, k), dtype=object)                                 # LM = cell(1, k) (:37)
        for i in range(k):
            lm[0, i] = np.asarray(LM[i], dtype=np.float64)
        ws.update({"LM": lm, "LM_transformed": _cell(evaluated["LM_transformed"]),
                   "LM_acc": np.asarray(evaluated["LM_acc"]).reshape(-1, 1),
                   "LM_transformed_acc": np.asarray(evaluated["LM_transformed_acc"]).reshape(-1, 1)})
    savemat(path, ws, format="5", oned_as="column")
    return path


def load_results(path):
    """reads a file written by save_results (or by the reference's demos with -v7 or older) into a dict"""
    from scipy.io import loadmat
    return loadmat(path, squeeze_me=False, struct_as_record=False)
