"""Policy export with the signature of the reference's SavedModel (src/gym/stable_solve.py:66-90):
one input "ob" (a batch of observations), two outputs "act" (the deterministic action, the mean of
the Gaussian head) and "stochastic_act" (a sample).  TensorFlow is not part of this stack, so the
container is TorchScript: `torch.jit.load(path)(ob) -> (act, stochastic_act)`; `load_policy` wraps
that as the `act(obs)` callable the plugin-side controller (shim.PolicyRateController, the
counterpart of src/udt-plugins/testing/loaded_agent.py) wants."""
import copy
import json
import os

import numpy as np
import torch
from torch import nn


class _Exported(nn.Module):
    def __init__(self, pi, log_std):
        super().__init__()
        self.pi = pi
        self.log_std = nn.Parameter(log_std.detach().clone(), requires_grad=False)

    def forward(self, ob: torch.Tensor):
        act = self.pi(ob)
        stochastic_act = act + torch.randn_like(act) * torch.exp(self.log_std)
        return act, stochastic_act


def export_policy(policy, export_dir, history_len=10, features=None):
    """Write <export_dir>/policy.pt (TorchScript, CPU) and signature.json naming inputs and outputs."""
    os.makedirs(export_dir, exist_ok=True)
    # a COPY of the network goes to the CPU: nn.Module.to() moves parameters in place, and the caller's policy (and its
    # optimizer state) must stay where it is
    mod = _Exported(copy.deepcopy(policy.pi), policy.log_std).to("cpu").eval()
    scripted = torch.jit.script(mod)
    path = os.path.join(export_dir, "policy.pt")
    scripted.save(path)
    obs_dim = policy.pi[0].in_features
    with open(os.path.join(export_dir, "signature.json"), "w") as f:
        json.dump({"inputs": {"ob": [None, obs_dim]}, "outputs": {"act": [None, 1], "stochastic_act": [None, 1]},
                   "history_len": history_len, "features": features, "format": "torchscript"}, f, indent=1)
    return path


def load_policy(export_dir, stochastic=False):
    """`act(obs) -> action` over numpy observations from an exported directory (or the .pt file)."""
    path = export_dir if export_dir.endswith(".pt") else os.path.join(export_dir, "policy.pt")
    mod = torch.jit.load(path, map_location="cpu")

    def act(obs):
        ob = torch.as_tensor(np.asarray(obs, dtype=np.float32)).reshape(1, -1) if np.ndim(obs) == 1 else \
            torch.as_tensor(np.asarray(obs, dtype=np.float32))
        with torch.no_grad():
            a, sa = mod(ob)
        out = (sa if stochastic else a).numpy()
        return out[0] if np.ndim(obs) == 1 else out

    return act
