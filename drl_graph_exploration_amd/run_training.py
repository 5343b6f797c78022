"""`run_training.py` of the reference (scripts/run_training.py): ONE training epoch - load the pickled trainer
(`saved_training.pkl`, replay buffer included) and the two model checkpoints, run `trainer.running(...)`, pickle the
trainer back.  `python -m drl_graph_exploration_amd.run_training <DQN|A2C> <GCN> [--data-root ../data] [--n-envs 64]`."""
import argparse
import os
import pickle

import torch


def run_epoch(training_method, model_name, data_root="../data", n_envs=64):
    from .train import make_models, paths, second_name
    case_path, object_path, _ = paths(data_root, training_method, model_name)
    full_file_name = os.path.join(object_path, "saved_training.pkl")
    with open(full_file_name, "rb") as f:
        trainer = pickle.load(f)
    device = torch.device("cuda", torch.cuda.current_device())
    m1, m2 = make_models(training_method, model_name, device)
    m1.load_state_dict(torch.load(os.path.join(object_path, "Model_Policy.pt"), map_location=device))
    m2.load_state_dict(torch.load(os.path.join(object_path, second_name(training_method)), map_location=device))
    trainer.running(m1, m2, n_envs=n_envs)
    with open(full_file_name, "wb") as f:
        pickle.dump(trainer, f)
    return trainer


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("training_method", choices=["DQN", "A2C"])
    ap.add_argument("model_name")
    ap.add_argument("--data-root", default="../data")
    ap.add_argument("--n-envs", type=int, default=64)
    args = ap.parse_args(argv)
    run_epoch(args.training_method, args.model_name, args.data_root, args.n_envs)


if __name__ == "__main__":
    main()
