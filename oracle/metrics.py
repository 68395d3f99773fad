"""Oracle (TEST INFRASTRUCTURE): integer metrics of the hot path, numpy/float64.

Restates, without sklearn:
* confusion_matrix(y_true, y_pred, labels=range(C))  -- call site nsga_penalty.py:355
  (rows = true class, cols = predicted class; labels outside range(C) ignored,
  as sklearn does when ``labels`` is given)
* calculate_fpr V1 ... nsga_penalty.py:351-364 (= sa_nsga_penalty.py:189-202,
                       mobo_penalty.py:203-216, init_sa_nsga_local.py:137-143)
* calculate_fpr V3 ... ablation_study/sa_nsga_local.py:138-141
* the y_true quirk ... nsga_penalty.py:387 (argmax over an (N,1) array == zeros)
* objective/CV assembly nsga_penalty.py:428-441
Pinned by tests/golden/fpr_golden.json / objectives_golden.json, which hold the
outputs of the reference's own functions executed in the build container.
"""
import numpy as np

FPR_V1 = 0        # macro mean over all C classes, 0.0 when FP+TN == 0
FPR_V1_QUIRK = 1  # V1 with y_true forced to all-zeros (nsga_penalty.py:387)
FPR_V3 = 2        # classes with total - rowsum == 0 are dropped from the mean


def confusion_matrix(y_true, y_pred, num_classes):
    y_true = np.asarray(y_true).astype(np.int64).ravel()
    y_pred = np.asarray(y_pred).astype(np.int64).ravel()
    ok = (y_true >= 0) & (y_true < num_classes) & (y_pred >= 0) & (y_pred < num_classes)
    cm = np.zeros((num_classes, num_classes), dtype=np.int64)
    np.add.at(cm, (y_true[ok], y_pred[ok]), 1)
    return cm


def fpr_from_confusion(cm, variant=FPR_V1):
    cm = np.asarray(cm, dtype=np.int64)
    C = cm.shape[0]
    total = int(cm.sum())
    vals = []
    for i in range(C):
        col = int(cm[:, i].sum())
        row = int(cm[i, :].sum())
        fp = col - int(cm[i, i])
        if variant == FPR_V3:
            den = total - row
            if den > 0:
                vals.append(fp / den)
        else:
            tn = total - (row + col - int(cm[i, i]))
            vals.append(fp / (fp + tn) if (fp + tn) > 0 else 0.0)
    if variant == FPR_V3:
        return float(np.mean(vals)) if vals else 0.0
    return float(np.mean(vals))


def calculate_fpr(y_true, y_pred, num_classes, variant=FPR_V1):
    y_true = np.asarray(y_true).ravel()
    if variant == FPR_V1_QUIRK:
        y_true = np.zeros_like(y_true)
        return fpr_from_confusion(confusion_matrix(y_true, y_pred, num_classes), FPR_V1)
    return fpr_from_confusion(confusion_matrix(y_true, y_pred, num_classes), variant)


def assemble(hparams, acc, size_mb, fpr, min_accuracy, max_model_size, max_fpr):
    """One result dict, nsga_penalty.py:428-441 (float64 python arithmetic)."""
    g1 = max(0.0, min_accuracy - acc)
    g2 = max(0.0, size_mb - max_model_size)
    g3 = max(0.0, fpr - max_fpr)
    return {"hparams": hparams, "objs": [-acc, size_mb, fpr], "CV": g1 + g2 + g3}
