"""Training entry point: `python -m video_gcp_amd.train --path <exp_dir> [flags]` — the counterpart of
/root/reference/gcp/prediction/train.py (ModelTrainer.run / train / train_epoch / val / save_checkpoint / resume, :24-213)
with the flags of gcp/prediction/training/gcp_builder.py:188-247 that concern the device path.

What is replaced: `output = model(inputs); losses = model.loss(...); total.backward(); optimizer.step()` (train.py:155-163)
runs as `GCPTrainStep.step` (HIP forward + explicit backward + RAdam, training.py); data-parallel training is one process
per GPU (launch with `python -m torch.distributed.run --nproc-per-node N -m video_gcp_amd.train ...`), every rank on its own
shard, gradients averaged with ONE RCCL all-reduce over the flat gradient vector (gcp_builder.py:71-78 used nn.DataParallel).
What is NOT here (out of the hot path): HDF5 datasets, TensorBoard logging, the control evaluation.  The
experiment directory holds `conf.json` ({"config": "c2", "overrides": {...}, "num_epochs": n, "lr": x, "batches_per_epoch": k});
`--feed_random_data 1` (the reference's own debug flag, gcp_builder.py:239) feeds the seeded synthetic batches of SURVEY §8d,
which is the only data source this build ships; `dataset` may also be any iterable of input dicts passed to ModelTrainer.
"""
import argparse
import json
import os
import time

import torch


def get_cmd_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--path", help="path to the experiment directory (conf.json; weights/ is created inside)")
    p.add_argument("--config", default=None, help="named configuration (c1..c5) when there is no conf.json")
    p.add_argument("--dont_save", default=False, type=int)
    p.add_argument("--resume", default="", type=str, help="'latest', an epoch number or a checkpoint path")
    p.add_argument("--train", default=True, type=int, help="if 0, runs one validation epoch")
    p.add_argument("--skip_first_val", default=False, type=int)
    p.add_argument("--gpu", default=-1, type=int)
    p.add_argument("--strict_weight_loading", default=True, type=int)
    p.add_argument("--deterministic", default=False, type=int)
    p.add_argument("--log_outputs_interval", default=10, type=int)
    p.add_argument("--feed_random_data", default=True, type=int)
    p.add_argument("--verbose_timing", default=False, type=int)
    p.add_argument("--num_epochs", default=None, type=int)
    p.add_argument("--batches_per_epoch", default=None, type=int)
    return p.parse_args(argv)


class SyntheticLoader:
    """Seeded synthetic batches: a different shard per rank, a different batch per step."""

    def __init__(self, hp, n_batches, seed, device, variant="B"):
        self.hp, self.n, self.seed, self.device, self.variant = hp, n_batches, seed, device, variant

    def __len__(self):
        return self.n

    def __iter__(self):
        from .synthetic import make_inputs, make_inputs_device
        on_gpu = torch.device(self.device).type == "cuda"
        for i in range(self.n):
            if on_gpu:                                   # drawn on the device: the host path is 9x slower than a training step
                yield make_inputs_device(self.hp, self.seed + i, self.variant, self.device)
            else:
                inputs, _, _ = make_inputs(self.hp, seed=self.seed + i, variant=self.variant)
                yield {k: v.to(self.device) for k, v in inputs.items()}


class ModelTrainer:
    def __init__(self, args=None, train_loader=None, val_loader=None):
        from . import dist as D
        from .hparams import config
        from .model import GCPTreeModel
        from .params import init_params
        from .training import GCPTrainStep
        self.cmd_args = args = args if args is not None else get_cmd_args()
        from .conf_loader import load_conf
        # <path>/conf.py in the reference's format (`configuration` + `model_config`, gcp_builder.py:129-147) or <path>/conf.json
        name = args.config or "c2"
        # an explicit --config wins over conf.json's "config" entry (and names the default experiment directory below)
        self.hp, conf, self.ignored_conf_keys = load_conf(args.path, default="c2", name=args.config)
        for k in self.ignored_conf_keys:
            if k.startswith("defaulted:"):
                print(f"[train] {k[10:]} is not in the configuration file (the reference reads it from the dataset spec): "
                      f"using {getattr(self.hp, k[10:])}", flush=True)
        hp = self.hp
        self.exp_path = args.path or os.path.join("experiments", name)
        self.rank, self.local_rank, self.world = D.init_from_env()
        if args.gpu >= 0:
            self.local_rank = args.gpu
        torch.cuda.set_device(self.local_rank)
        self.device = torch.device("cuda", self.local_rank)
        self.num_epochs = args.num_epochs if args.num_epochs is not None else conf.get("num_epochs", 1)
        nb = args.batches_per_epoch if args.batches_per_epoch is not None else conf.get("batches_per_epoch", 10)
        seed = 0 if args.deterministic else conf.get("seed", 0)
        # same initial weights on every rank (same seed) = the broadcast of DataParallel replicas ...
        # configuration['model'] (gcp_builder.py:75) chooses the class: TreeModel or the flat VRNN baseline SequentialModel
        # (experiments/prediction/base_configs/gcp_sequential.py)
        self.model_kind = conf.get("model", "tree")
        if self.model_kind == "sequential":
            from .params import init_params_sequential
            from .sequential import GCPSequentialModel
            from .training_sequential import SequentialTrainStep
            model_cls, init_fn, step_cls = GCPSequentialModel, init_params_sequential, SequentialTrainStep
        else:
            model_cls, init_fn, step_cls = GCPTreeModel, init_params, GCPTrainStep
        self.model = model_cls(hp, params=init_fn(hp, seed=seed), device=self.device)
        # ... but a DIFFERENT stream of latent noise / auxiliary-model index draws per rank, as nn.DataParallel's replicas drew
        # different numbers for their different shards (gcp_builder.py:71-78); reproducible through `seed` / --deterministic
        torch.cuda.manual_seed(seed * max(self.world, 1) + self.rank + 1)
        self.model.reseed()                   # (a second trainer of the process seeded with the same value starts the same stream again)
        pg = torch.distributed.group.WORLD if self.world > 1 else None
        lr = conf.get("lr") if conf.get("lr") is not None else 1e-3
        self.trainer = step_cls(self.model, lr=lr, betas=(conf.get("adam_beta", 0.9), 0.999), process_group=pg,
                                optimizer=conf.get("optimizer", "radam"), momentum=conf.get("momentum", 0),
                                gradient_clip=conf.get("gradient_clip"))                   # gcp_builder.py:174-186,255-263
        if not args.feed_random_data and train_loader is None:
            raise ValueError("no dataset reader ships with this build: pass --feed_random_data 1 or give ModelTrainer a loader")
        self.train_loader = train_loader or SyntheticLoader(hp, nb, 1000 + 100000 * self.rank, self.device)
        self.val_loader = val_loader or SyntheticLoader(hp, max(1, nb // 5), 500000 + 100000 * self.rank, self.device)
        self.global_step = 0
        self.log = []

    # ---- train.py:28-54 ----
    def run(self):
        start_epoch = 0
        if self.cmd_args.resume:
            start_epoch = self.resume(self.cmd_args.resume)
        if self.cmd_args.train:
            self.train(start_epoch)
        else:
            self.val()

    # ---- train.py:56-70, checkpoint_handler.py:31-74 ----
    def resume(self, ckpt, path=None):
        from . import checkpoint as CK
        folder = os.path.join(self.exp_path if path is None else path, "weights")
        try:
            f = CK.get_resume_ckpt_file(ckpt, folder)
        except CK.NoCheckpointsException:               # train.py:59-62: `--resume latest` on a fresh experiment starts at epoch 0
            return 0
        self.global_step, epoch, opt = CK.load_weights(f, self.model, strict=bool(self.cmd_args.strict_weight_loading))
        if opt is not None:
            self.trainer.load_optimizer_state(opt)
        # schedules follow the step counter (the KL-weight burn-in, base_gcp.py:121-128, counts model.step() calls)
        self.model.n_steps = int(self.global_step)
        self.model._set_kl_weight()
        self.model.train()
        return epoch + 1

    # ---- train.py:95-104 ----
    def train(self, start_epoch):
        if not self.cmd_args.skip_first_val:
            self.val()
        for epoch in range(start_epoch, self.num_epochs):
            self.train_epoch(epoch)
            if not self.cmd_args.dont_save and self.rank == 0:
                self.save_checkpoint(epoch)
            self.val()

    # ---- train.py:106-115 ----
    def save_checkpoint(self, epoch):
        from . import checkpoint as CK
        return CK.save_checkpoint(self.model, os.path.join(self.exp_path, "weights"), epoch, self.global_step,
                                  self.trainer.optimizer_state())

    # ---- train.py:132-193 ----
    def train_epoch(self, epoch):
        self.model.train()
        end = time.time()
        for batch_idx, inputs in enumerate(self.train_loader):
            out = self.trainer.step(inputs)            # zero_grad, forward, loss, backward, (all-reduce), optimizer.step
            self.model.step()                          # train.py:163
            if self.global_step % self.cmd_args.log_outputs_interval == 0:
                total = float(out.raw["losses"][5])    # the only host sync of the loop, on logging steps
                self.log.append((self.global_step, total))
                if self.rank == 0:
                    print("itr: {} Train Epoch: {} [{}/{}]\tLoss: {:.6f}\t{:.3f}s/batch".format(
                        self.global_step, epoch, batch_idx, len(self.train_loader), total,
                        (time.time() - end) / max(1, self.cmd_args.log_outputs_interval)), flush=True)
                end = time.time()
            self.global_step += 1

    # ---- train.py:195-238: prior-sampled prediction (val_mode) + the training forward for the NLL, no gradient ----
    def val(self):
        tot, n = 0.0, 0
        start = time.time()
        for inputs in self.val_loader:
            with self.model.val_mode(pred_length=False):
                self.model(inputs, "test")
            out = self.model(inputs)
            losses = self.model.loss(inputs, out)
            tot += float(self.model.get_total_loss(inputs, losses).value)
            n += 1
        avg = tot / max(n, 1)
        self.last_val = avg
        if self.rank == 0:
            print("\nTest set: Average loss: {:.4f} in {:.2f}s\n".format(avg, time.time() - start), flush=True)
        return avg


if __name__ == "__main__":
    ModelTrainer().run()
