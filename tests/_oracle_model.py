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

    def clip_gradients(self, clip_norm):
        norm = torch.linalg.vector_norm(self.grad_arena)
        self.grad_arena.mul_(clip_norm / torch.clamp(norm, min=clip_norm))

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


class BucketedOracleModel(OracleModel):
    """OracleModel with the engine's DATA-PARALLEL surface: `set_grad_callback(fn, stream)` and gradient buckets
    reported one by one DURING forward_backward, in the engine's order (head, cross layers L-1..0, audio stack,
    motion stack - include/fact_hip.h fact_set_grad_callback).  A bucket's arena range is only written right
    before its callback; everything not yet reported holds NaN, so a reducer that touches a range early, twice
    or out of order poisons the result."""

    def __init__(self, cfg, seed=0):
        super().__init__(cfg, seed)
        self._cb = None
        off, self.offsets = 0, {}
        for n, sz in zip(self.names, self.sizes):
            self.offsets[n] = (off, sz)
            off += sz
        L = cfg["cross"]["layers"]

        def rng(pred):
            idx = [i for i, n in enumerate(self.names) if pred(n)]
            lo = self.offsets[self.names[idx[0]]][0]
            hi = sum(self.offsets[self.names[idx[-1]]])
            assert idx == list(range(idx[0], idx[-1] + 1))  # contiguous in Keras variable order
            return lo, hi - lo
        self.buckets = [rng(lambda n: n.startswith("cross_modal_layer/output/"))]
        for l in range(L - 1, -1, -1):
            self.buckets.append(rng(lambda n, l=l: n.startswith("cross_modal_layer/transformer/layer_%d/" % l)))
        self.buckets.append(rng(lambda n: n.startswith("audio_")))
        self.buckets.append(rng(lambda n: n.startswith("motion_")))
        assert sum(c for _, c in self.buckets) == self.grad_arena.numel()

    def set_grad_callback(self, fn, comm_stream):
        self._cb = fn

    def forward_backward(self, inputs, target, loss_scale=1.0):
        loss, grads, _ = O.loss_and_grads(self.params, self.cfg, inputs["motion_input"], inputs["audio_input"],
                                          target, num_replicas=1.0 / loss_scale)
        flat = torch.cat([grads[n].flatten() for n in self.names])
        if self._cb is None:
            self.grad_arena += flat
            return loss
        assert float(self.grad_arena.abs().sum()) == 0.0  # zeroed by the optimizer step
        self.grad_arena.fill_(float("nan"))
        for b, (off, cnt) in enumerate(self.buckets):
            self.grad_arena[off:off + cnt] = flat[off:off + cnt]
            self._cb(b, off, cnt)
        return loss

    def cast_bucket_to_bf16(self, src, dst, stream=None):
        dst.copy_(src.to(torch.bfloat16))

    def cast_bucket_from_bf16(self, src, dst, stream=None):
        dst.copy_(src.to(dst.dtype))
