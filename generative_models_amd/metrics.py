"""Sample-quality metrics of the heavy evaluation, written from their definitions:

* Fréchet distance between two Gaussians fitted to latent sets — the quantity the reference logs as `eval/fid`
  (gms/common.py:267-288).  d = m(mu_a, mu_b) + tr(S_a) + tr(S_b) - 2 tr((S_a S_b)^(1/2)), with the reference's one deviation
  from the textbook formula kept: m is the MEAN of the squared mean differences, not their sum.
* k-NN manifold precision / recall / F1 (Kynkäänniemi et al., arXiv 1904.06991; gms/common.py:291-319): a point is covered by a
  set if it lies strictly inside the k-th-neighbour ball of at least one member of the set.
* `eval_heavy`: the sampling loop of gms/main.py:95-149 that feeds them (same metric keys, `y = -1` = unconditional).

Everything here works on [N, Z] latents of a feature extractor passed in by the caller (the reference's pretrained
autoencoder / classifier TorchScript files are not part of its checkout).  Off the throughput path: 500 x 64 latents."""
import numpy as np
import torch
import torch.nn.functional as F


def _trace_sqrt_product(cov_a, cov_b):
    """tr((A B)^(1/2)) for covariance matrices: the principal square root has the square roots of the eigenvalues of A B as
    its eigenvalues, so its trace is their sum (complex arithmetic: round-off can push a zero eigenvalue below zero)."""
    eig = np.linalg.eigvals(cov_a @ cov_b).astype(np.complex128)
    return np.sqrt(eig).sum()


def compute_fid(x, y):
    """x, y: [N, Z] numpy latents -> float; NaN for anything that cannot be evaluated (wrong rank, too few rows, a failed
    eigen-decomposition) — the reference swallows every failure into NaN too."""
    x, y = np.asarray(x), np.asarray(y)
    if x.ndim != 2 or y.ndim != 2 or x.shape[1] != y.shape[1] or min(x.shape[0], y.shape[0]) < 2:
        return np.nan
    try:
        mean_term = np.mean(np.square(x.mean(axis=0) - y.mean(axis=0)))
        cov_x = np.atleast_2d(np.cov(x, rowvar=False))
        cov_y = np.atleast_2d(np.cov(y, rowvar=False))
        value = mean_term + np.trace(cov_x) + np.trace(cov_y) - 2.0 * _trace_sqrt_product(cov_x, cov_y)
        return float(np.real(value))
    except (np.linalg.LinAlgError, ValueError, FloatingPointError):
        return np.nan


def knn_radii(points, k):
    """Distance from every point to its k-th nearest OTHER point of the same set ([N] tensor)."""
    pairwise = torch.cdist(points, points)
    return pairwise.kthvalue(k + 1, dim=1).values            # position 0 of the sorted row is the point itself (distance 0)


def coverage(manifold, queries, k=3):
    """Fraction of `queries` lying strictly inside at least one k-NN ball of `manifold`."""
    inside = torch.cdist(queries, manifold) < knn_radii(manifold, k)[None, :]
    return inside.any(dim=1).float().mean()


def precision_recall_f1(*, real, gen, k=3):
    precision = coverage(real, gen, k)            # generated samples that fall on the data manifold
    recall = coverage(gen, real, k)               # data points the generated manifold reaches
    return {"precision": precision, "recall": recall, "f1": 2 * (precision * recall) / (precision + recall)}


class _LatentBank:
    """Named lists of latent batches, concatenated on demand."""

    def __init__(self):
        self._rows = {}

    def add(self, name, z):
        self._rows.setdefault(name, []).append(z.detach().float())

    def get(self, name):
        return torch.cat(self._rows[name])


@torch.inference_mode()
def eval_heavy(logger, model, test_ds, autoencoder, classifier, G, total_samples=500):
    """Draw at least `total_samples` samples batch by batch (unconditional: y = -1; additionally class-conditional when
    G.class_cond), embed samples and test images, log eval/fid, eval/precision|recall|f1 (+ cond_* and classifier_loss).
    The reference also logs ignite's own FID (`ignite_fid`); pytorch-ignite is not installed here, so that key is absent."""
    bank = _LatentBank()
    clf_losses = []
    drawn = 0
    for x, y in ((b[0].to(G.device), b[1].to(G.device)) for b in test_ds):
        n = x.shape[0]
        if G.class_cond:
            guided = model.sample(n, y=y)
            clf_losses.append(F.cross_entropy(classifier(guided), y).item())
            bank.add("cond", autoencoder(guided))
        bank.add("gen", autoencoder(model.sample(n, y=torch.full_like(y, -1))))
        bank.add("real", autoencoder(x))
        drawn += n
        if drawn >= total_samples:
            break
    real = bank.get("real")

    def against_real(name):
        gen = bank.get(name)
        scores = precision_recall_f1(real=real, gen=gen)
        scores["fid"] = compute_fid(gen.cpu().numpy(), real.cpu().numpy())
        return scores

    # The reference's metric keys (`eval/fid`, `eval/precision` ...) are kept for values taken in the REFERENCE's feature space (its
    # TorchScript arbiters).  With the stand-ins of arbiters.py the same quantities are logged as `eval/randfeat_*` /
    # `eval/centroid_classifier_loss`, so that a log or hps file can never pass them off as reference-space numbers.
    space = "randfeat_" if getattr(autoencoder, "stand_in", False) else ""
    results = {space + key: val for key, val in against_real("gen").items()}
    if G.class_cond:
        results[("centroid_" if getattr(classifier, "stand_in", False) else "") + "classifier_loss"] = clf_losses
        results.update({space + "cond_" + key: val for key, val in against_real("cond").items()})
    for key, val in results.items():
        val = val.detach().cpu().numpy() if isinstance(val, torch.Tensor) else val
        logger[f"eval/{key}"] += [float(np.mean(val))]
    return results
