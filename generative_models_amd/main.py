"""Training driver — mirror of reference gms/main.py (load_model_and_data :43-92, train :152-217).

Same flag surface (`DG`, :20-40, merged with the model's `DG`, two-pass argparse with prefix matching), same
eval-first epoch order and the same calls on the model (to / eval / loss / evaluate / save / train / train_step),
same metric key naming (`<model>/train/<key>`, `<model>/test/<key>`, `dt/train`, `num_vars`).

Differences forced by the environment (no network, no tensorboard/ignite in the image): the data source is a
synthetic MNIST-shaped generator (`--data synthetic`: x fp32 [B,1,28,28] in [-1,1] with the 2x-1 scaling and
zero pad-to-32 of gms/common.py:104-112, y int64 in 0..9), the writer is `common.NullWriter`, and the heavy eval
(FID / precision-recall, :95-149) is SURVEY §8f "next" and skipped.  The per-step `.cpu()` of :215 is deferred to
the end of the epoch so the training loop never synchronises with the device.

    python -m generative_models_amd.main --model=diffusion --epochs=1 --bs 32
    torchrun --nproc-per-node 8 -m generative_models_amd.main --model=diffusion   (one process per GPU, RCCL)
"""
import argparse
import os
import time
from itertools import count
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import yaml

from . import common, parallel
from . import metrics as heavy

DG = common.AttrDict()     # gms/main.py:20-40
DG.model = "vae"
DG.bs = 64
DG.hidden_size = 256
DG.device = "cuda"
DG.epochs = 50
DG.save_n = 5
DG.logdir = Path("./logs/")
DG.lr = 3e-4
DG.class_cond = 0
DG.binarize = 1
DG.pad32 = 0
DG.mode = "train"
DG.weights_from = Path(".")
DG.autoencoder = Path("./weights/autoencoder.pt")
DG.classifier = Path("./weights/classifier.pt")
DG.eval_heavy = 0
DG.skip_training = 0
# additions
DG.data = "synthetic"      # 'synthetic' | 'mnist' (IDX files under <data_root>/MNIST/raw)
DG.data_root = Path("data")
DG.train_batches = 8       # synthetic batches per epoch (per rank)
DG.test_batches = 2


class SyntheticMNIST:
    """MNIST-shaped batches: ~85 % background pixels at exactly -1 (after 2x-1), the rest uniform in [-1, 1]."""

    def __init__(self, bs, n_batches, pad32, binarize, device, seed):
        self.bs, self.n, self.pad32, self.binarize, self.device = bs, n_batches, pad32, binarize, device
        self.gen = torch.Generator().manual_seed(seed)
        self._seed, self._ctr = int(seed), 0

    def __iter__(self):
        on_gpu = str(self.device).startswith("cuda")
        for _ in range(self.n):
            if on_gpu:      # drawn on the device (Philox kernels): the host generator costs 20+ ms per batch and would bound the loop
                from . import ops
                shape = (self.bs, 1, 28, 28)
                nq = (self.bs * 784 + 3) // 4
                raw = ops.rng_uniform(shape, self._seed, self._ctr, self.device)
                ink = ops.rng_uniform(shape, self._seed, self._ctr + nq, self.device) < 0.15
                y = (ops.rng_uniform((self.bs,), self._seed, self._ctr + 2 * nq, self.device) * 10).long().clamp_(0, 9)
                self._ctr += 2 * nq + (self.bs + 3) // 4
            else:
                raw = torch.rand((self.bs, 1, 28, 28), generator=self.gen)
                ink = torch.rand((self.bs, 1, 28, 28), generator=self.gen) < 0.15
                y = torch.randint(0, 10, (self.bs,), generator=self.gen)
            x = torch.where(ink, raw, torch.zeros_like(raw))
            x = (x > 0.5).float() if self.binarize else 2 * x - 1          # gms/common.py:105-109
            if self.pad32:
                x = torch.nn.functional.pad(x, (2, 2, 2, 2))                # :110-111 pads with 0
            yield x, y

    def __len__(self):
        return self.n


def init_distributed():
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
        backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend=backend)


def load_model_and_data(argv=None):
    parser = argparse.ArgumentParser()
    for key, value in DG.items():
        parser.add_argument(f"--{key}", type=common.args_type(value), default=value)
    tempG, _ = parser.parse_known_args(argv)

    defaults = {}
    if tempG.weights_from != Path("."):
        with open(tempG.weights_from.parent / "hps.yaml") as f:
            loadedG = common.AttrDict(yaml.load(f, Loader=yaml.Loader))
        for key, value in loadedG.items():
            defaults[key] = value
            if key not in tempG:
                parser.add_argument(f"--{key}", type=type(value), default=value)
        Model = common.discover_models()[loadedG.model]
    else:
        Model = common.discover_models()[tempG.model]
        for key, value in Model.DG.items():
            defaults[key] = value
            if key not in tempG:
                parser.add_argument(f"--{key}", type=type(value), default=value)
        defaults["logdir"] = tempG.logdir / tempG.model
    defaults.pop("full_cmd", None)
    parser.set_defaults(**defaults)
    G = common.AttrDict(parser.parse_args(argv).__dict__)
    init_distributed()
    if G.device == "cuda" and torch.cuda.is_available() and dist.is_initialized():
        G.device = f"cuda:{torch.cuda.current_device()}"
    model = Model(G=G).to(G.device)
    if G.weights_from != Path("."):
        model.load_state_dict(torch.load(G.weights_from, map_location=G.device), strict=False)
    if parallel.world() > 1 and hasattr(model, "net"):
        parallel.GradSync(model.net).broadcast_params(0)
    r = parallel.rank()
    if G.data == "mnist":         # gms/main.py:84 load_mnist: the IDX files under data/MNIST/raw (cannot be downloaded here)
        from . import data as mnist_data
        train_ds, test_ds = mnist_data.load_mnist(G.bs, G.binarize, G.pad32, root=str(G.data_root), device=G.device, seed=1000,
                                                  rank=r, world=parallel.world())
    elif G.data == "synthetic":
        train_ds = SyntheticMNIST(G.bs, G.train_batches, G.pad32, G.binarize, G.device, seed=1000 + r)
        test_ds = SyntheticMNIST(G.bs, G.test_batches, G.pad32, G.binarize, G.device, seed=2000 + r)
    else:
        raise ValueError(f"--data {G.data!r}: 'synthetic' or 'mnist'")
    print("num_vars", common.count_vars(model))
    # heavy eval (gms/main.py:86-91): TorchScript feature extractors; the reference's weight files are not in the checkout
    # (.MISSING_LARGE_BLOBS), so a missing file is an error only when --eval_heavy asks for it
    autoencoder = classifier = None
    if G.eval_heavy:
        need = [Path(G.autoencoder)] + ([Path(G.classifier)] if G.get("class_cond", 0) else [])
        missing = [str(f) for f in need if not f.exists()]
        if missing:       # DiffusionModel.DG.eval_heavy defaults to 1 as in the reference; without the files it cannot run
            print(f"eval_heavy disabled: {', '.join(missing)} not found (the reference's weight files are not in its checkout)")
            G.eval_heavy = 0
        else:
            autoencoder = torch.jit.load(str(G.autoencoder)).to(G.device)
            if G.get("class_cond", 0):
                classifier = torch.jit.load(str(G.classifier)).to(G.device)
    return model, train_ds, test_ds, autoencoder, classifier, G


def train(model, train_ds, test_ds, autoencoder, classifier, G):
    writer = common.NullWriter(G.logdir)
    logger = common.dump_logger({}, writer, 0, G)
    for epoch in count(0):
        # TEST (eval first, gms/main.py:159-183)
        model.eval()
        with torch.no_grad():
            if hasattr(model, "loss"):
                for test_batch in test_ds:
                    test_x, test_y = test_batch[0].to(G.device), test_batch[1].to(G.device)
                    _, test_metrics = model.loss(test_x, test_y)
                    for key in test_metrics:
                        prefix_key = f"{G.model}/test/{key}" if not key == "nlogp" else f"eval/{key}"
                        logger[prefix_key] += [test_metrics[key].detach().cpu().item()]
            else:
                test_batch = next(iter(test_ds))
                test_x, test_y = test_batch[0].to(G.device), test_batch[1].to(G.device)
            eval_time = time.time()
            model.evaluate(writer, test_x, test_y, epoch)
            logger["dt/eval"] = time.time() - eval_time
        logger["num_vars"] = common.count_vars(model)
        if epoch % G.save_n == 0 and parallel.rank() == 0:
            Path(G.logdir).mkdir(parents=True, exist_ok=True)
            model.save(Path(G.logdir), test_x, test_y)
            print("SAVED MODEL", G.logdir)
            if G.eval_heavy:                       # gms/main.py:191-196
                print("RUNNING HEAVY EVAL...")
                eval_heavy_time = time.time()
                heavy.eval_heavy(logger, model, test_ds, autoencoder, classifier, G)
                logger["dt/eval_heavy"] = time.time() - eval_heavy_time
                print("DONE HEAVY EVAL")
        logger = common.dump_logger(logger, writer, epoch, G)
        if epoch >= G.epochs:
            break
        # TRAIN (gms/main.py:202-217)
        model.train()
        train_time = time.time()
        pending = []
        for batch in train_ds:
            if G.skip_training:
                break
            train_x, train_y = batch[0].to(G.device), batch[1].to(G.device)
            metrics = model.train_step(train_x, train_y)
            pending.append(metrics)
        for metrics in pending:       # one device->host transfer pass per epoch instead of one sync per step
            for key in metrics:
                prefix_key = f"{G.model}/train/{key}" if not key == "nlogp" else f"train/{key}"
                logger[prefix_key] += [metrics[key].detach().cpu()]
        logger["dt/train"] = time.time() - train_time
    return logger, writer


if __name__ == "__main__":
    model, train_ds, test_ds, autoencoder, classifier, G = load_model_and_data()
    train(model, train_ds, test_ds, autoencoder, classifier, G)
