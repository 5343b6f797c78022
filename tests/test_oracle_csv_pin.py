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


def pinned_plan(env, acts, ks, choice):
    """The action list of one pinned decision on the oracle env: frontier index, [variant, frontier index] (a line
    plan whose length is a multiple of the edge length up to one ulp, remainder on the other side) or a [gx, gy] goal."""
    if isinstance(choice, int):
        return acts[ks + choice]
    if isinstance(choice[0], str):
        plan = list(acts[ks + choice[1]])
        return plan[:-1] if choice[0] == "no_tail" else plan + [(0.0, 0.0, 0.0)]
    return env._sim.line_plan((choice[0], choice[1]))


exact_rows = {}  # seed -> rows whose entropy was asserted at 1e-9 (test_entropy_is_exact_away_from_decision_boundaries)


# every seed the oracle follows for >= 10 rows; 4 runs the reference's whole episode (179 actions, 184 poses),
# 5 / 30 / 38 / 48 follow it for 95-136 rows
LONG = [4, 5, 30, 38, 48]


def _pinned_seeds():
    """EVERY seed of the fixture with at least one pinned row (44 of the CSV's 50: the same set the GPU replay covers)."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "csv_pin.json")) as f:
        return sorted(int(k) for k, v in json.load(f)["seeds"].items() if len(v["rows"]) > 0)


@pytest.mark.parametrize("lo", _pinned_seeds())
def test_oracle_tracks_reference_csv(pins, dqn_weights, lo):
    pin = pins["seeds"][str(lo)]
    rows_ref = np.array(pin["rows"])
    assert len(rows_ref) >= 1
    env = O.OracleEnv(40, lo)
    st = 0
    agree = n_frontier_choices = n_exact = 0
    for d, (choice, gcn_choice) in enumerate(zip(pin["choices"], pin["gcn_choices"])):
        A, X, _, fro = env.graph_matrix()
        acts = env.actions_all_goals()
        ks = A.shape[0] - fro
        if gcn_choice >= 0:  # (-1: decisions found by the goal search, the network's pick was not recorded)
            ei, ea, x = O.data_process(A, X)
            with torch.no_grad():
                q = gcn_ref.gcn_forward(dqn_weights, torch.tensor(x), torch.tensor(ei), torch.tensor(ea)).view(-1).numpy()
            assert int(np.argmax(q[-fro:])) == gcn_choice
        if isinstance(choice, int):
            n_frontier_choices += 1
            agree += int(gcn_choice == choice)
        plan = pinned_plan(env, acts, ks, choice)
        # the fixture's executed actions are this plan (cut where the episode ends)
        np.testing.assert_allclose(np.array(plan)[:len(pin["plans"][d])], np.array(pin["plans"][d]), rtol=0, atol=1e-12)
        for a in plan[:len(pin["plans"][d])]:
            obs, _, _ = env.step(a)
            got = np.array([env.get_landmark_error(), O.map_entropy(obs), env.max_uncertainty_of_trajectory()])
            ref = rows_ref[st]
            # landmark error and max pose-covariance trace: 1e-4 relative (observed <= 1e-5, often 1e-9..1e-16)
            assert got[0] == pytest.approx(ref[0], rel=1e-4)
            assert got[2] == pytest.approx(ref[2], rel=1e-4)
            # map entropy: a cell whose range / bearing test is decided by less than the reference's own truncation of the
            # back-substitution (iSAM2 wildfire threshold 1e-3, not restated) can fall on the other side there, and such flips
            # accumulate along an episode: 2 % for those maps.  A map in which NO cell of NO pose is that close to a boundary
            # must reproduce the reference's entropy to round-off (observed: 2e-16 on all 201 such rows of the fixture).
            if env._sim.knife_edge_cells(1e-3).any():
                assert got[1] == pytest.approx(ref[1], rel=2e-2)
            else:
                assert got[1] == pytest.approx(ref[1], rel=1e-9)
                n_exact += 1
            st += 1
    assert st == len(rows_ref)
    exact_rows[lo] = n_exact
    # (the oracle's map drifts from the reference's through knife-edge cells; seeds pinned for only a few decisions are exempt)
    assert agree >= 0.85 * n_frontier_choices or n_frontier_choices < 8


def test_entropy_is_exact_away_from_decision_boundaries():
    """Runs after the parametrised replay: the tight entropy assertion was not vacuous."""
    if len(exact_rows) < 10:
        pytest.skip("the replay tests were deselected")
    assert sum(exact_rows.values()) >= 60


def test_pin_covers_long_trajectories(pins):
    """The fixture reaches the trajectory lengths the reference's episodes have: >= 2000 pinned rows in total, a whole
    179-action episode and four more seeds beyond 90 actions (i.e. 100-184 poses with the 5 of reset())."""
    seeds = pins["seeds"]
    assert sum(len(v["rows"]) for v in seeds.values()) >= 2000
    assert seeds["4"]["finished"] and len(seeds["4"]["rows"]) == seeds["4"]["episode_rows"] == 179
    assert sum(1 for v in seeds.values() if len(v["rows"]) > 90) >= 5
    for v in seeds.values():
        assert sum(len(p) for p in v["plans"]) == len(v["rows"]) and len(v["plans"]) == len(v["choices"]) == len(v["goals"])


def test_gcn_restatement_agrees_with_reference_choices(pins):
    pairs = [(a, b) for v in pins["seeds"].values() for a, b in zip(v["choices"], v["gcn_choices"]) if isinstance(a, int)]
    same = sum(int(a == b) for a, b in pairs)
    assert len(pairs) > 400 and same >= len(pairs) - 10
