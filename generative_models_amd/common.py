"""Plugin surface shared by models and the driver — mirror of the parts of reference gms/common.py that the
diffusion hot path sits behind: `AttrDict` (:24-26), `convert_camel_to_snake` (:33-35),
`discover_models` (:38-55), `args_type` (:85-92), `count_vars` (:95-96), `GM` (:138-174), `dump_logger` (:65-82).

The evaluation / visualisation utilities of the reference file (FID, precision-recall, tensorboard grids, the
MNIST loader) are outside the hot path and are not reproduced here.
"""
import importlib
import re
import subprocess
import sys
from collections import defaultdict
from pathlib import Path

import numpy as np
import torch
import yaml
from torch import nn


class AttrDict(dict):
    __setattr__ = dict.__setitem__
    __getattr__ = dict.__getitem__


def prefix_dict(name, d):
    return {name + key: d[key] for key in d}


_WORD_STARTS = re.compile(r"(?<=.)(?=[A-Z][a-z])|(?<=[a-z0-9])(?=[A-Z])")


def convert_camel_to_snake(name):
    """'DiffusionModel' -> 'diffusion_model', 'VQVAE' -> 'vqvae', 'RNNModel' -> 'rnn_model': the registry-key rule of gms/common.py:33-35
    (an underscore in front of every capital that starts a lower-case word or follows a lower-case letter / digit)."""
    return _WORD_STARTS.sub("_", name).lower()


class GM(nn.Module):
    """GenerativeModel base class: the contract the training driver relies on (gms/common.py:138-174)."""

    DG = AttrDict()   # per-model default flags; every key becomes a --flag of the driver

    def __init__(self, G):
        super().__init__()
        self.G = G
        self.optimizer = None

    def save(self, path, test_x=None, test_y=None):
        torch.save(self.state_dict(), Path(path) / "model.pt")

    def _default_optimizer(self):
        if self.optimizer is None:
            self.optimizer = torch.optim.Adam(self.parameters(), self.G.lr)
        return self.optimizer

    def train_step(self, x, y):
        """Default step for plugins that only define `loss(x, y) -> (loss, metrics)` (gms/common.py:158-169): Adam on all parameters,
        created on first use."""
        if not hasattr(self, "loss"):
            raise AssertionError("the default train_step needs a `loss(x, y)` method that returns (loss, metrics)")
        opt = self._default_optimizer()
        opt.zero_grad()
        objective, metrics = self.loss(x, y)
        objective.backward()
        opt.step()
        return metrics

    def evaluate(self, writer, x, y, epoch):
        assert False, "you need to implement the evaluate method. make some samples or something."


_NOT_PLUGIN_FILES = ("__init__", "main", "common")          # the reference's filter on file names (gms/common.py:45)


def _plugin_modules(package):
    """Dotted names of the package's modules that may define a plugin: file name passes the reference's filter AND the source
    contains a class deriving from something called *GM* (so that discovery does not import kernels / bindings needlessly)."""
    root = Path(importlib.import_module(package).__file__).parent
    for path in sorted(root.rglob("*.py")):
        if any(word in path.name for word in _NOT_PLUGIN_FILES):
            continue
        try:
            source = path.read_text()
        except OSError:
            continue
        if re.search(r"^\s*class\s+\w+\(.*GM\w*\)", source, flags=re.M):
            yield ".".join(path.relative_to(root.parent).with_suffix("").parts)


def discover_models(package="generative_models_amd"):
    """{snake_case(class name): class} for every GM subclass defined in the package (gms/common.py:38-55).  The reference CLI
    spelling `--model=diffusion` (run_all.sh:16) is registered as an alias of `diffusion_model`."""
    found = {}
    for modname in _plugin_modules(package):
        for attr, obj in vars(importlib.import_module(modname)).items():
            if isinstance(obj, type) and obj is not GM and issubclass(obj, GM):
                found[convert_camel_to_snake(attr)] = obj
    if "diffusion_model" in found:
        found["diffusion"] = found["diffusion_model"]
    return found


def to_numpy(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x


def _parse_bool(text):
    return bool(("False", "True").index(text))          # anything else is a ValueError, as in the reference


def _parse_int(text):
    return float(text) if ("e" in text or "." in text) else int(text)


def args_type(default):
    """Flag parser chosen from the default's type (gms/common.py:85-92): bools are the words 'True' / 'False', ints also accept
    '1e3'-style floats, Paths are expanded, everything else is parsed by its own type."""
    for kind, parser in ((bool, _parse_bool), (int, _parse_int), (Path, lambda text: Path(text).expanduser())):
        if isinstance(default, kind):
            return parser
    return type(default)


def count_vars(module):
    """Parameter count as the reference reports it (gms/common.py:95-96); zero padding a module keeps around its parameters (the narrow
    U-Net widths, `padding_numel`) is not counted."""
    total = sum(int(np.prod(p.shape)) for p in module.parameters())
    return total - sum(int(getattr(m, "padding_numel", 0)) for m in module.modules())


class NullWriter:
    """Stand-in for tensorboard's SummaryWriter (not installed in this image): keeps the call surface, stores scalars."""

    def __init__(self, logdir=None):
        self.logdir = logdir
        self.scalars = defaultdict(list)

    def add_scalar(self, key, val, step):
        self.scalars[key].append((step, float(val)))

    def add_image(self, *a, **k):
        pass

    def add_video(self, *a, **k):
        pass

    def flush(self):
        pass


def write_grid(writer, tag, x, epoch):
    """25 samples as one 5x5 image (gms/common.py:177-180: '(n1 n2) c h w -> c (n1 h) (n2 w)')."""
    assert tuple(x.shape) == (25, 1, 28, 28)
    grid = x.reshape(5, 5, 1, 28, 28).permute(2, 0, 3, 1, 4).reshape(1, 140, 140)
    writer.add_image(tag, grid, epoch)
    return grid


def write_gridvid(writer, tag, x, epoch):
    """A [T, 25, 1, 28, 28] trajectory as a 3-channel 5x5-grid video, fps = min(T // 3, 60) (gms/common.py:183-193)."""
    T = x.shape[0]
    assert tuple(x.shape[1:]) == (25, 1, 28, 28)
    vid = x.reshape(T, 5, 5, 1, 28, 28).permute(0, 3, 1, 4, 2, 5).reshape(T, 1, 140, 140)[None]
    vid = vid.repeat(1, 1, 3, 1, 1)
    writer.add_video(tag, vid, epoch, fps=min(T // 3, 60))
    return vid


def _git_head():
    try:
        return subprocess.check_output(["git", "rev-parse", "HEAD"], stderr=subprocess.DEVNULL).decode("ascii").strip()
    except Exception:
        return "unknown"       # the reference requires a checkout; outside one the field is still written


def dump_logger(logger, writer, i, G):
    """End-of-epoch report (gms/common.py:65-82): the mean of every logged series goes to the writer and to stdout, the flags (with
    the command line and the git hash) to <logdir>/hps.yaml; returns a fresh logger."""
    rule = "=" * 30
    means = {key: np.mean(series) for key, series in logger.items()}
    for key, val in means.items():
        writer.add_scalar(key, val, i)
    G.full_cmd = "python " + " ".join(sys.argv)
    G.commit_hash = _git_head()
    print("\n".join([rule, str(i)] + [f"{key} {val}" for key, val in means.items()] + [G.full_cmd]))
    logdir = Path(G.logdir)
    logdir.mkdir(parents=True, exist_ok=True)
    (logdir / "hps.yaml").write_text(yaml.dump(dict(G), width=float("inf")))
    print(rule)
    writer.flush()
    return defaultdict(list)
