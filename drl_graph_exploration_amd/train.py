"""`train.py` of the reference (scripts/train.py) over the drlgx trainers: creates the trainer, pickles it to
`<data_root>/training_object_data/<case>/saved_training.pkl`, saves the initial `Model_Policy.pt` / `Model_Target.pt`
(`Model_Value.pt` for A2C), then runs EXPLORE / epoch training epochs - each one is `run_training.run_epoch` (the
reference spawns `python3 run_training.py <method> <model>` per epoch; `--subprocess` does the same) - and after each
epoch appends `temp_reward.csv` / `temp_loss.csv` to the TensorBoard log `<data_root>/torch_logs/<case>/` under the tags
'Train/avg_reward' and 'Train/loss' (scripts/train.py:85-94).

    python -m drl_graph_exploration_amd.train [DQN|A2C] [GCN] [--data-root ../data] [--epochs N] [--n-envs 64] [--subprocess]
"""
import argparse
import os
import pickle
import subprocess
import sys

import numpy as np
import torch

from . import networks
from .policy import A2C, DeepQ
from .tfevents import SummaryWriter


def paths(data_root, training_method, model_name):
    case_path = training_method + "_" + model_name + "/"
    return case_path, os.path.join(data_root, "training_object_data", case_path), os.path.join(data_root, "torch_logs", case_path)


def make_models(training_method, model_name, device):
    if model_name != "GCN":
        raise NotImplementedError("only the GCN models are on the accelerated path (GG-NN / g-U-Net: SURVEY.md, out of scope)")
    if training_method == "DQN":
        return networks.GCN().to(device), networks.GCN().to(device)
    if training_method == "A2C":
        return networks.PolicyGCN().to(device), networks.ValueGCN().to(device)
    raise ValueError(training_method)


def second_name(training_method):
    return "Model_Target.pt" if training_method == "DQN" else "Model_Value.pt"


def log_epoch(writer, object_path):
    """scripts/train.py:85-94."""
    for name, tag in (("temp_reward.csv", "Train/avg_reward"), ("temp_loss.csv", "Train/loss")):
        data = np.loadtxt(os.path.join(object_path, name), delimiter=",", ndmin=2)
        for j in range(np.shape(data)[0]):
            writer.add_scalar(tag, data[j][1], data[j][0])
    writer.flush()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("training_method", nargs="?", default="DQN", choices=["DQN", "A2C"])
    ap.add_argument("model_name", nargs="?", default="GCN")
    ap.add_argument("--data-root", default="../data")
    ap.add_argument("--epochs", type=int, default=None, help="default: EXPLORE / epoch like the reference")
    ap.add_argument("--epoch-steps", type=int, default=None, help="override the trainer's `epoch` (environment steps per epoch)")
    ap.add_argument("--observe", type=int, default=None, help="override the trainer's OBSERVE")
    ap.add_argument("--batch", type=int, default=None, help="override the DQN minibatch size (reference: 64)")
    ap.add_argument("--n-envs", type=int, default=64)
    ap.add_argument("--subprocess", action="store_true", help="one `python -m ...run_training` process per epoch, like the reference")
    args = ap.parse_args(argv)
    from . import run_training
    case_path, object_path, log_path = paths(args.data_root, args.training_method, args.model_name)
    os.makedirs(object_path, exist_ok=True)
    writer = SummaryWriter(log_dir=log_path)
    trainer = DeepQ(case_path, args.model_name, data_root=args.data_root) if args.training_method == "DQN" else \
        A2C(case_path, data_root=args.data_root)
    if args.epoch_steps is not None:
        trainer.epoch = args.epoch_steps
    if args.observe is not None and hasattr(trainer, "OBSERVE"):
        trainer.OBSERVE = args.observe
    if args.batch is not None and hasattr(trainer, "BATCH"):
        trainer.BATCH = args.batch
    epoch_nums = trainer.EXPLORE / trainer.epoch if args.epochs is None else args.epochs
    with open(os.path.join(object_path, "saved_training.pkl"), "wb") as f:
        pickle.dump(trainer, f)
    device = torch.device("cuda", torch.cuda.current_device())
    m1, m2 = make_models(args.training_method, args.model_name, device)
    torch.save(m1.state_dict(), os.path.join(object_path, "Model_Policy.pt"))
    torch.save(m2.state_dict(), os.path.join(object_path, second_name(args.training_method)))
    for _ in range(int(epoch_nums)):
        if args.subprocess:
            subprocess.check_call([sys.executable, "-m", "drl_graph_exploration_amd.run_training", args.training_method, args.model_name,
                                   "--data-root", args.data_root, "--n-envs", str(args.n_envs)])
        else:
            run_training.run_epoch(args.training_method, args.model_name, args.data_root, args.n_envs)
        log_epoch(writer, object_path)
    writer.close()
    return writer.path


if __name__ == "__main__":
    main()
