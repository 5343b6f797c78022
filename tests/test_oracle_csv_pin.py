"""CPU test: the oracle replayed end-to-end against the reference's shipped evaluation CSV
(data/test_result/40_DQN_GCN.csv; fixture tests/golden/csv_pin.json made by
scripts/make_csv_pin_fixture.py).  This is the pin that ties the oracle (RNG streams, hash order,
geometry, iSAM2 linearisation policy, marginals, occupancy, graph export, GCN forward) to the
reference's own outputs."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import gcn_ref
from oracle import oracle as O


@pytest.fixture(scope="module")
def pins(golden_dir):
    return json.load(open(os.path.join(golden_dir, "csv_pin.json")))


@pytest.fixture(scope="module")
def dqn_weights(golden_dir):
    return torch.load(os.path.join(golden_dir, "DQN_GCN_MyModel.pt"), map_location="cpu")


def test_state_dict_layout(dqn_weights):
    shapes = {k: tuple(v.shape) for k, v in dqn_weights.items()}
    assert shapes == {"conv1.weight": (5, 1000), "conv1.bias": (1000,), "conv2.weight": (1000, 1000),
                      "conv2.bias": (1000,), "fully_con1.weight": (1, 1000), "fully_con1.bias": (1,)}


@pytest.mark.parametrize("lo", [0, 1, 2, 3, 8, 13, 16, 24, 39, 48])
def test_oracle_tracks_reference_csv(pins, dqn_weights, lo):
    pin = pins["seeds"][str(lo)]
    rows_ref = np.array(pin["rows"])
    assert len(rows_ref) >= 19
    env = O.OracleEnv(40, lo)
    st = 0
    agree = 0
    for choice, gcn_choice in zip(pin["choices"], pin["gcn_choices"]):
        A, X, _, fro = env.graph_matrix()
        ei, ea, x = O.data_process(A, X)
        with torch.no_grad():
            q = gcn_ref.gcn_forward(dqn_weights, torch.tensor(x), torch.tensor(ei), torch.tensor(ea)).view(-1).numpy()
        assert int(np.argmax(q[-fro:])) == gcn_choice
        agree += int(gcn_choice == choice)
        acts = env.actions_all_goals()
        ks = A.shape[0] - fro
        for a in acts[ks + choice]:
            obs, _, _ = env.step(a)
            got = np.array([env.get_landmark_error(), O.map_entropy(obs), env.max_uncertainty_of_trajectory()])
            ref = rows_ref[st]
            # landmark error and max pose-covariance trace: 1e-4 relative (observed <= 1e-5, often 1e-9..1e-16)
            assert got[0] == pytest.approx(ref[0], rel=1e-4)
            assert got[2] == pytest.approx(ref[2], rel=1e-4)
            # map entropy: single boundary cells may flip (DESIGN.md "oracle pin"): <= 0.5 % here
            assert got[1] == pytest.approx(ref[1], rel=5e-3)
            st += 1
    assert st == len(rows_ref)
    assert agree >= len(pin["choices"]) - 1


def test_gcn_restatement_agrees_with_reference_choices(pins):
    tot = sum(len(v["choices"]) for v in pins["seeds"].values())
    same = sum(int(a == b) for v in pins["seeds"].values() for a, b in zip(v["choices"], v["gcn_choices"]))
    assert tot > 200 and same >= tot - 3
