"""CPU tests of the on-disk formats of the reference's training scripts: TensorBoard event files
(scripts/train.py:22-23, :85-94) against the head of the reference's own shipped log (tests/golden/A2C_GCN_events_head.*,
made from data/torch_logs/A2C_GCN/ by the snippet in this file's docstring of `test_reference_log_head`), and the pickled
trainer / replay hand-off (scripts/train.py:33-35, scripts/run_training.py:15-16, :63-64)."""
import json
import os
import pickle
import struct

import numpy as np
import pytest
import torch

from drl_graph_exploration_amd import tfevents as T


def test_crc32c_known_answers():
    assert T.crc32c(b"123456789") == 0xE3069283  # the CRC-32C check value
    assert T.crc32c(b"") == 0
    assert T.masked_crc(struct.pack("<Q", 24)) == 0x224B7FA3  # header CRC of the reference log's first record (24 bytes)


def test_reference_log_head(golden_dir):
    """The first 64 records of the reference's A2C_GCN TensorBoard log: every CRC verifies, the decoded scalars are the
    recorded ones, and re-encoding each scalar with this writer reproduces the reference's record bytes exactly.
    (fixture: `buf[:end of record 64]` of data/torch_logs/A2C_GCN/events.out.tfevents.1581732701.pc.21830.0)"""
    path = os.path.join(golden_dir, "A2C_GCN_events_head.tfevents")
    want = json.load(open(os.path.join(golden_dir, "A2C_GCN_events_head.json")))["scalars"]
    got = T.read_scalars(path, check_crc=True)
    assert len(got) == len(want) == 63
    for (wall, step, tag, val), (s, t, v) in zip(got, want):
        assert (step, tag) == (s, t) and val == pytest.approx(v, rel=1e-7)
    assert got[0][1:] == (1100, "Train/avg_reward", pytest.approx(-0.41841381788253784))
    recs = list(T.read_records(path))
    assert recs[0][9:] == b"\x1a\rbrain.Event:2" and T.encode_version_event(struct.unpack("<d", recs[0][1:9])[0]) == recs[0]
    for rec, (wall, step, tag, val) in zip(recs[1:], got):
        assert T.encode_scalar_event(tag, val, step, wall) == rec


def test_summary_writer_round_trip(tmp_path):
    with T.SummaryWriter(log_dir=str(tmp_path / "logs" / "DQN_GCN")) as w:
        for k in range(50):
            w.add_scalar("Train/avg_reward", -0.5 + 0.01 * k, 100 * k)
            w.add_scalar("Train/loss", 1.0 / (k + 1), 64 * k + 1)
        path = w.path
    assert os.path.basename(path).startswith("events.out.tfevents.")
    sc = T.read_scalars(path)
    assert len(sc) == 100 and sc[0][1:3] == (0, "Train/avg_reward") and sc[-1][1:3] == (64 * 49 + 1, "Train/loss")
    assert sc[-1][3] == pytest.approx(1.0 / 50, rel=1e-6)
    # a flipped byte is detected
    raw = bytearray(open(path, "rb").read())
    raw[60] ^= 1
    open(path, "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        T.read_scalars(path)


def test_trainer_pickle_round_trip(tmp_path):
    """saved_training.pkl: the trainer with its replay buffer survives pickle (compact host graphs, scalars, counters) and
    trains on after the reload."""
    from drl_graph_exploration_amd.networks import GraphData
    from drl_graph_exploration_amd.policy import A2C, DeepQ
    rng = np.random.RandomState(0)
    dq = DeepQ("t/", "GCN", data_root=str(tmp_path))
    dq.BATCH, dq.step_t, dq.epsilon = 4, 1234, 0.77

    def graph(n):
        e = rng.randint(0, n, (2, 2 * n))
        return GraphData(torch.randn(n, 5), torch.as_tensor(e), torch.rand(2 * n))
    for _ in range(6):
        n = int(rng.randint(4, 9))
        dq.buffer.append((graph(n), n - 1, float(rng.randn()), graph(n + 1), False, 2))
    dq.total_reward = np.append(dq.total_reward, [0.1, -0.2])
    blob = pickle.dumps(dq)
    dq2 = pickle.loads(blob)
    assert (dq2.step_t, dq2.epsilon, dq2.BATCH, len(dq2.buffer)) == (1234, 0.77, 4, 6)
    np.testing.assert_array_equal(dq2.total_reward, dq.total_reward)
    for a, b in zip(dq.buffer, dq2.buffer):
        assert torch.equal(a[0].x, b[0].x) and torch.equal(a[3].edge_index, b[3].edge_index) and a[1:3] == b[1:3] and a[4:] == b[4:]

    class Lin(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.l = torch.nn.Linear(5, 1)

        def forward(self, data, prob, batch=None):
            return self.l(data.x)
    pol, tgt = Lin(), Lin()
    opt = torch.optim.Adam(pol.parameters(), lr=1e-3)
    dq2._train_minibatch(torch.device("cpu"), pol, tgt, opt)
    assert dq2.temp_loss > 0
    a2c = A2C("t2/", data_root=str(tmp_path))
    a2c.step_t = 80
    a2c2 = pickle.loads(pickle.dumps(a2c))
    assert a2c2.step_t == 80 and a2c2.nstep == 40 and len(a2c2.buffer) == 0
