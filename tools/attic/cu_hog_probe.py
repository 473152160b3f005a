"""De-risking the 8-GPU run on one GPU: what does the train step lose when N CUs are held by another
kernel for the whole step - the situation a communication library's persistent channel workgroups create, since every
big-tile GEMM workgroup needs a whole CU?  A spin kernel (fact_debug_cu_hog: one workgroup per CU) holds N CUs on its own
stream while normal train steps run.  Prints ms/step per N (interleaved rounds)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mint_amd import _lib as L
from mint_amd import configs, model_builder
from mint_amd.trainer import Adam, SingleTaskTrainer


def main():
    torch.cuda.set_device(0)
    lib = L.lib()
    pipe = configs.fact_v5_deeper_t10_cm12()
    model = model_builder.build(pipe.multi_modal_model, True)
    B = 16
    g = torch.Generator().manual_seed(1)
    batch = {"motion_input": torch.randn(B, 120, 225, generator=g).cuda(), "audio_input": torch.randn(B, 240, 35, generator=g).cuda(),
             "target": torch.randn(B, 20, 225, generator=g).cuda()}
    model.build(B, 225, 35)
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        model.set_option(k, int(v))

    class Rep:
        def __iter__(self):
            return self

        def __next__(self):
            return batch
    tr = SingleTaskTrainer(Rep(), "target", model, optimizer=Adam(1e-4))
    it = iter(Rep())
    hog = torch.cuda.Stream()
    for _ in range(5):
        tr.train_step(it)
    torch.cuda.synchronize()
    res = {}
    for rnd in range(2):
        for n in [int(x) for x in os.environ.get("HOG_NS", "0,4,8,16,32,64").split(",")]:
            steps = 10
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if n:
                L.check(lib.fact_debug_cu_hog(n, 9000 * steps + 3000, C.c_void_p(hog.cuda_stream)))
            e0.record()
            for _ in range(steps):
                tr.train_step(it)
            e1.record()
            e1.synchronize()
            res.setdefault(n, []).append(e0.elapsed_time(e1) / steps)
            torch.cuda.synchronize()
    print("cu_hog_probe %s: " % " ".join(sys.argv[1:]) + "  ".join("N=%d %s" % (n, "/".join("%.3f" % x for x in v)) for n, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
