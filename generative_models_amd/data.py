"""MNIST input path (reference gms/common.py:102-132 `load_mnist`) without torchvision: an IDX reader over the files
torchvision's `MNIST('data', download=True)` leaves under data/MNIST/raw/, the reference's transform chain and its
DataLoader settings (shuffle, drop_last).  There is no network here, so the files have to be present already; the
driver's default stays `--data synthetic`.

Transform chain, in the reference's order (gms/common.py:104-111):
    ToTensor()               uint8 HxW -> float32 [1, H, W] / 255
    binarize:  (x > 0.5).float()          else:  x.float(); 2 * x - 1
    pad32:     F.pad(x, (2, 2, 2, 2))     (zeros — also for the [-1, 1] data, whose background is -1)
"""
import gzip
import os
import struct

import numpy as np
import torch

_IDX_DTYPES = {0x08: np.uint8, 0x09: np.int8, 0x0B: ">i2", 0x0C: ">i4", 0x0D: ">f4", 0x0E: ">f8"}
FILES = {True: ("train-images-idx3-ubyte", "train-labels-idx1-ubyte"),
         False: ("t10k-images-idx3-ubyte", "t10k-labels-idx1-ubyte")}


def read_idx(path):
    """IDX file (optionally .gz) -> numpy array.  Header: 2 zero bytes, dtype code, ndim, then ndim big-endian uint32 sizes."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        raw = f.read()
    zero, code, ndim = struct.unpack(">HBB", raw[:4])
    if zero != 0 or code not in _IDX_DTYPES:
        raise ValueError(f"{path}: not an IDX file (magic {raw[:4]!r})")
    dims = struct.unpack(">" + "I" * ndim, raw[4:4 + 4 * ndim])
    arr = np.frombuffer(raw, dtype=_IDX_DTYPES[code], offset=4 + 4 * ndim)
    if arr.size != int(np.prod(dims)):
        raise ValueError(f"{path}: header says {dims}, payload has {arr.size} items")
    return arr.reshape(dims)


def write_idx(path, arr):
    """Inverse of read_idx for uint8 arrays (fixtures, tests)."""
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "wb") as f:
        f.write(struct.pack(">HBB", 0, 0x08, arr.ndim) + struct.pack(">" + "I" * arr.ndim, *arr.shape) + arr.tobytes())


def _find(root, name):
    for cand in (os.path.join(root, "MNIST", "raw", name), os.path.join(root, "MNIST", "raw", name + ".gz"),
                 os.path.join(root, name), os.path.join(root, name + ".gz")):
        if os.path.exists(cand):
            return cand
    raise FileNotFoundError(f"{name}[.gz] not found under {root}/MNIST/raw (MNIST cannot be downloaded here: no network); "
                            f"use --data synthetic or place the four IDX files there")


def transform(images_u8, binarize=True, pad32=False):
    """uint8 [N, 28, 28] -> float32 [N, 1, H, W]; gms/common.py:104-111."""
    x = torch.from_numpy(np.array(images_u8, dtype=np.uint8, copy=True)).unsqueeze(1).to(torch.float32).div(255)    # ToTensor()
    if binarize:
        x = (x > 0.5).float()
    else:
        x = 2 * x.float() - 1
    if pad32:
        x = torch.nn.functional.pad(x, (2, 2, 2, 2))
    return x


class MnistLoader:
    """One split as an iterable of (x, y) batches: shuffled every epoch, last partial batch dropped (gms/common.py:116-131),
    the transform applied once up front, batches copied to `device` from pinned memory."""

    def __init__(self, root, train, bs, binarize=True, pad32=False, device="cpu", seed=0, rank=0, world=1):
        img_name, lab_name = FILES[bool(train)]
        images = read_idx(_find(root, img_name))
        labels = read_idx(_find(root, lab_name))
        if images.ndim != 3 or labels.ndim != 1 or images.shape[0] != labels.shape[0]:
            raise ValueError(f"unexpected MNIST shapes {images.shape} / {labels.shape}")
        self.x = transform(images, binarize, pad32)
        self.y = torch.from_numpy(labels.astype(np.int64))
        self.bs, self.device = int(bs), device
        self.rank, self.world = rank, world
        self.gen = torch.Generator().manual_seed(seed)
        if torch.cuda.is_available() and str(device).startswith("cuda"):
            self.x, self.y = self.x.pin_memory(), self.y.pin_memory()

    def __len__(self):
        return (self.x.shape[0] // self.world) // self.bs

    def __iter__(self):
        perm = torch.randperm(self.x.shape[0], generator=self.gen)
        perm = perm[self.rank::self.world]                       # data parallel: disjoint shards of one shared permutation
        for i in range(len(self)):
            idx = perm[i * self.bs:(i + 1) * self.bs]
            yield self.x[idx].to(self.device, non_blocking=True), self.y[idx].to(self.device, non_blocking=True)


def load_mnist(bs, binarize=True, pad32=False, root="data", device="cpu", seed=0, rank=0, world=1):
    """-> (train_loader, test_loader), the call shape of gms/common.py:102."""
    return (MnistLoader(root, True, bs, binarize, pad32, device, seed, rank, world),
            MnistLoader(root, False, bs, binarize, pad32, device, seed + 1, rank, world))


class SyntheticMNIST:
    """MNIST-shaped stand-in for `load_mnist` when the IDX files are absent (no network here): per epoch `n_batches`
    batches of (x fp32 [bs, 1, 28|32, 28|32], y int64 [bs] in 0..9).  About 85 % of the raw pixels are background (0 before the
    transform chain), the rest uniform in [0, 1); the reference's transform chain is then applied as `transform` does
    (binarize or 2x-1, zero pad32).  On a GPU the draws come from the device's counter-based Philox kernels — a host generator
    costs 20+ ms per 1024-image batch and would bound the training loop."""

    def __init__(self, bs, n_batches, pad32, binarize, device, seed):
        self.bs, self.n_batches = int(bs), int(n_batches)
        self.pad32, self.binarize, self.device = bool(pad32), bool(binarize), device
        self.seed, self._counter = int(seed), 0
        self._host_gen = torch.Generator().manual_seed(self.seed)

    def __len__(self):
        return self.n_batches

    def _draw(self):
        shape = (self.bs, 1, 28, 28)
        if str(self.device).startswith("cuda"):
            from . import ops
            quads = (self.bs * 784 + 3) // 4                     # Philox counters one uniform tensor of this shape consumes
            base = self._counter
            self._counter += 2 * quads + (self.bs + 3) // 4
            raw = ops.rng_uniform(shape, self.seed, base, self.device)
            ink = ops.rng_uniform(shape, self.seed, base + quads, self.device) < 0.15
            labels = (ops.rng_uniform((self.bs,), self.seed, base + 2 * quads, self.device) * 10).long().clamp_(0, 9)
            return raw, ink, labels
        raw = torch.rand(shape, generator=self._host_gen)
        ink = torch.rand(shape, generator=self._host_gen) < 0.15
        return raw, ink, torch.randint(0, 10, (self.bs,), generator=self._host_gen)

    def __iter__(self):
        for _ in range(self.n_batches):
            raw, ink, labels = self._draw()
            x = raw * ink                                          # background pixels are exactly 0 before the transform
            x = (x > 0.5).float() if self.binarize else 2 * x - 1  # gms/common.py:105-109
            if self.pad32:
                x = torch.nn.functional.pad(x, (2, 2, 2, 2))       # :110-111 (zeros)
            yield x, labels
