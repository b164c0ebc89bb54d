"""Heavy-eval metrics of the reference driver: `compute_fid` (gms/common.py:267-288) and
`precision_recall_f1` (gms/common.py:291-319) on latent vectors, and the `eval_heavy` loop (gms/main.py:95-149).

They work on [N, Z] latents of a feature extractor (the reference's pretrained autoencoder / classifier, whose weight
files are not part of the checkout — `.MISSING_LARGE_BLOBS`); any callables can be passed.  Off the throughput path:
500 x 64 latents, stock numpy / torch ops."""
import numpy as np
import torch
import torch.nn.functional as F
from scipy.linalg import fractional_matrix_power


def compute_fid(x, y):
    """Fréchet distance between Gaussians fitted to x and y ([N, Z] numpy).  As in the reference: the mean term is the MEAN
    (not the sum) of squared differences, the real part is returned, any failure gives NaN (gms/common.py:273-288)."""
    try:
        assert x.ndim == 2 and y.ndim == 2
        pmu, tmu = np.mean(x, 0), np.mean(y, 0)
        pcov, tcov = np.cov(x, rowvar=False), np.cov(y, rowvar=False)
        assert pcov.shape[0] == x.shape[-1]
        fid = np.mean((pmu - tmu) ** 2) + np.trace(pcov + tcov - 2 * fractional_matrix_power(pcov.dot(tcov), 0.5))
        return fid.real
    except Exception:
        return np.nan


def _manifold_estimate(set_a, set_b, k=3):
    """Fraction of set_b inside the k-NN manifold of set_a (arXiv 1904.06991; gms/common.py:307-314)."""
    d = torch.cdist(set_a, set_a)
    radii = torch.topk(d, k + 1, largest=False).values[..., -1:]
    d2 = torch.cdist(set_a, set_b)
    return (d2 < radii).any(0).float().mean()


def precision_recall_f1(*, real, gen, k=3):
    precision = _manifold_estimate(real, gen, k)
    recall = _manifold_estimate(gen, real, k)
    f1 = 2 * (precision * recall) / (precision + recall)
    return {"precision": precision, "recall": recall, "f1": f1}


@torch.inference_mode()
def eval_heavy(logger, model, test_ds, autoencoder, classifier, G, total_samples=500):
    """gms/main.py:95-149.  Draws >= total_samples samples (unconditional: y = -1; class-conditional when G.class_cond),
    embeds samples and test images with `autoencoder`, logs eval/fid, eval/precision|recall|f1 (+ cond_*, classifier_loss).
    The reference's extra `ignite_fid` needs pytorch-ignite, which is not installed: it is logged only if importable."""
    from . import common
    sample_ct = 0
    all_z_sample, all_z_real, all_z_cond_sample = [], [], []
    metrics = {}
    if G.class_cond:
        metrics["classifier_loss"] = []
    for test_batch in test_ds:
        test_x, test_y = test_batch[0].to(G.device), test_batch[1].to(G.device)
        bs = test_x.shape[0]
        if G.class_cond:
            cond_samp = model.sample(bs, y=test_y)
            metrics["classifier_loss"].append(F.cross_entropy(classifier(cond_samp), test_y).item())
            all_z_cond_sample.append(autoencoder(cond_samp))
        samp = model.sample(bs, y=-torch.ones_like(test_y))
        all_z_real.append(autoencoder(test_x))
        all_z_sample.append(autoencoder(samp))
        sample_ct += bs
        if sample_ct >= total_samples:
            break
    z_samp, z_real = torch.cat(all_z_sample).float(), torch.cat(all_z_real).float()
    metrics["fid"] = compute_fid(z_samp.cpu().numpy(), z_real.cpu().numpy())
    metrics.update(precision_recall_f1(real=z_real, gen=z_samp))
    if G.class_cond:
        z_cond = torch.cat(all_z_cond_sample).float()
        cond = precision_recall_f1(real=z_real, gen=z_cond)
        cond["fid"] = compute_fid(z_cond.cpu().numpy(), z_real.cpu().numpy())
        metrics.update(common.prefix_dict("cond_", cond))
    for key, val in metrics.items():
        logger[f"eval/{key}"] += [np.mean(common.to_numpy(val))]
    return metrics
