"""Shared helpers for the parity tests."""
import numpy as np
import torch

from pychain_amd.graph import ChainGraph, ChainGraphBatch

GRAPH_FIELDS = ["forward_transitions", "forward_transition_probs", "forward_transition_indices",
                "backward_transitions", "backward_transition_probs", "backward_transition_indices",
                "final_probs", "initial_probs", "leaky_probs"]


def graph_from_npz(z, prefix):
    """Rebuild a ChainGraph from the arrays `graph_arrays` stored in a fixture."""
    t = {f: torch.from_numpy(np.array(z[prefix + f])) for f in GRAPH_FIELDS if prefix + f in z.files}
    return ChainGraph.from_tensors(
        t["forward_transitions"], t["forward_transition_probs"], t["forward_transition_indices"],
        t["backward_transitions"], t["backward_transition_probs"], t["backward_transition_indices"],
        t["final_probs"], t["initial_probs"], t.get("leaky_probs"),
        start_state=int(z[prefix + "start_state"]), log_domain=bool(z[prefix + "log_domain"]))


class RawBatch(object):
    """A ChainGraphBatch-like bag of batched tensors read from a fixture."""
    shared_graph = None

    def __init__(self, z, prefix):
        for f in GRAPH_FIELDS + ["start_state"]:
            if prefix + f in z.files:
                setattr(self, f, torch.from_numpy(np.array(z[prefix + f])))
            else:
                setattr(self, f, None)
        self.num_states = int(z[prefix + "num_states"])
        self.batch_size = int(z[prefix + "batch_size"])
        self.log_domain = bool(z[prefix + "log_domain"])


def batch_from_npz(z, prefix):
    rb = RawBatch(z, prefix)
    gb = ChainGraphBatch.__new__(ChainGraphBatch)
    gb.__dict__.update(rb.__dict__)
    gb.shared_graph = None
    gb._device_cache = {}
    return gb


def rel_err(a, b):
    """max |a-b| / max |b|  (the survey's gradient metric)."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b).max()
    return d / max(np.abs(b).max(), 1e-30)


class OracleChainLoss(torch.nn.Module):
    """ChainLoss(avg=False) evaluated by the CPU oracle, with autograd: the TEST stand-in for the per-rank loss where
    there is no GPU (tests/test_parallel.py, examples/train_tdnn.py --loss-cls helpers:OracleChainLoss).  What is
    under test there is the collective / DDP wiring around it."""

    def __init__(self, den_graph, leaky, avg=False):
        super().__init__()
        self.den_graph, self.leaky = den_graph, leaky

    def forward(self, x, lengths, num_graphs):
        import oracle as orc

        class F(torch.autograd.Function):
            @staticmethod
            def forward(ctx, xx):
                loss, grad = orc.chain_loss(xx, lengths, self.den_graph, num_graphs, self.leaky, avg=False)
                ctx.save_for_backward(torch.from_numpy(np.asarray(grad, dtype=np.float32)))
                return torch.tensor(float(loss))

            @staticmethod
            def backward(ctx, g):
                return ctx.saved_tensors[0] * g
        return F.apply(x)
