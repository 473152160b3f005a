"""CPU stand-in for mint_amd.fact_model.FACTModel used ONLY by the distributed host-logic tests:
same training surface (forward_backward / apply_adam / grad_arena), math by the oracle."""
import torch

from oracle import fact_oracle as O


class OracleModel:
    def __init__(self, cfg, seed=0):
        self.cfg = cfg
        self.params = O.init_params(cfg, seed=seed)
        self.names = [n for n, _ in O.param_shapes(cfg)]
        self.sizes = [self.params[n].numel() for n in self.names]
        self.grad_arena = torch.zeros(sum(self.sizes), dtype=torch.float64)
        self.m = {k: torch.zeros_like(v) for k, v in self.params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.params.items()}
        self.global_step = 0
        self.losses = []

    def forward_backward(self, inputs, target, loss_scale=1.0):
        loss, grads, _ = O.loss_and_grads(self.params, self.cfg, inputs["motion_input"], inputs["audio_input"],
                                          target, num_replicas=1.0 / loss_scale)
        self.grad_arena += torch.cat([grads[n].flatten() for n in self.names])
        return loss

    def apply_adam(self, lr, beta_1=0.9, beta_2=0.999, epsilon=1e-7, clip_norm=0.0):
        grads, off = {}, 0
        for n, s in zip(self.names, self.sizes):
            grads[n] = self.grad_arena[off:off + s].view_as(self.params[n]).clone()
            off += s
        self.params, self.m, self.v = O.adam_update(self.params, grads, self.m, self.v, self.global_step, lr,
                                                    beta_1, beta_2, epsilon, clip_norm)
        self.grad_arena.zero_()
        self.global_step += 1

    def flat_params(self):
        return torch.cat([self.params[n].flatten() for n in self.names])
