"""Probe for VERDICT r2 item 4 (forward co-runner): does the chip run TWO half-batch chains (B = 8 each, two engine
handles on two streams) faster than ONE B = 16 chain?  Upper bound of what intra-GPU half-batch chains could buy,
measured without touching the engine.  Usage: python tools/halfbatch_probe.py [fwd|train] [aux=0/1] [side=0/1]
Prints ms per (B = 16 worth of) work for: one handle B16; two handles B8 back-to-back on one stream; two handles B8 on
two streams."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import configs, model_builder


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
    aux = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    side = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    torch.cuda.set_device(0)
    pipe = configs.fact_v5_deeper_t10_cm12()

    def mk(B, seed):
        m = model_builder.build(pipe.multi_modal_model, True)
        g = torch.Generator().manual_seed(seed)
        batch = {"motion_input": torch.randn(B, 120, 225, generator=g).cuda(),
                 "audio_input": torch.randn(B, 240, 35, generator=g).cuda()}
        tgt = torch.randn(B, 20, 225, generator=g).cuda()
        m.build(B, 225, 35)
        m.set_option("aux_stream", aux)
        m.set_option("side_stream", side)
        return m, batch, tgt

    big = mk(16, 0)
    h0, h1 = mk(8, 1), mk(8, 2)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()

    def run(h):
        m, b, t = h
        if mode == "fwd":
            m(b)
        else:
            m.forward_backward(b, t)

    def one():
        run(big)

    def serial():
        run(h0); run(h1)

    def conc():
        cur = torch.cuda.current_stream()
        s0.wait_stream(cur); s1.wait_stream(cur)
        with torch.cuda.stream(s0):
            run(h0)
        with torch.cuda.stream(s1):
            run(h1)
        cur.wait_stream(s0); cur.wait_stream(s1)

    def time_ms(fn, iters=20):
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / iters

    res = {}
    for rnd in range(3):  # interleaved rounds
        for name, fn in (("one_B16", one), ("two_B8_serial", serial), ("two_B8_concurrent", conc)):
            res.setdefault(name, []).append(time_ms(fn))
    print("halfbatch_probe mode=%s aux=%d side=%d queues=%s: " % (mode, aux, side, os.environ.get("GPU_MAX_HW_QUEUES", "dflt"))
          + "  ".join("%s %s" % (k, "/".join("%.3f" % x for x in v)) for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
