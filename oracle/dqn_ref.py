"""TEST INFRASTRUCTURE (oracle): numpy restatement of the reference's DQN target loop and loss
(scripts/policy.py:152-177 target computation, :234-253 cost / train).  Imported only by tests/; the product
(drl_graph_exploration_amd/policy.py) never imports this module.

The quirk this keeps on purpose: `start_p` advances by the CURRENT-state node count
(`node_space = len(act)`), while `readout_j1_batch` is the target network's output over the collated NEXT states."""
import numpy as np


def reference_targets(acts, rewards, terminals, action_spaces, readout_j1_batch, gamma):
    """acts: list of one-hot action arrays over each sample's current-state nodes (minibatch[i][1]);
    rewards / terminals / action_spaces: minibatch[i][2] / [4] / [5]; readout_j1_batch: [sum of next-state nodes, 1]
    array (the `.cpu().detach().numpy()` of the target network's output).  Returns (a_batch, y_batch) float64."""
    a_batch = np.array([])
    y_batch = np.array([])
    start_p = 0
    for i in range(0, len(acts)):
        terminal = terminals[i]
        action_space = action_spaces[i]
        act = acts[i]
        a_batch = np.append(a_batch, act)
        node_space = len(act)
        temp_y = np.zeros(node_space)
        index = np.argmax(act)
        if terminal:
            temp_y[index] = rewards[i]
        else:
            temp_range = readout_j1_batch[start_p:start_p + node_space]
            temp_range = temp_range[-action_space:]
            # np.float32 scalar: under the reference's NumPy 1.x (legacy promotion) `python float * np.float32` is
            # float64; NumPy >= 2 (NEP 50) would keep float32 - the reference-era arithmetic is stated explicitly
            max_q = float(np.max(temp_range))
            temp_y[index] = rewards[i] + gamma * max_q
        start_p += node_space
        y_batch = np.append(y_batch, temp_y)
    return a_batch, y_batch


def reference_cost(pred, target, action, batch):
    """policy.py:234-239 with numpy float64 `target` / `action` and a float32 `pred` (the product promotes to float64)."""
    readout_action = pred.reshape(-1).astype(np.float64) * action
    return float(np.power(readout_action - target.reshape(-1), 2).sum() / batch)
