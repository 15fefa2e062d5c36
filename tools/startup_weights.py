"""Start-up cost of the network weights, N processes on one GPU box (round-5 review item 5: 3.5 - 4.7 s per rank at eight ranks).

    python tools/startup_weights.py --procs 8 [--legacy]

Every process maps the synthetic float32 checkpoint (tools/make_synthetic_checkpoint.py) with torch.load(mmap=True) and builds
NetWeights on cuda:0 -- round 6: upload float32, round / fold on the device (csrc/cpx_weights.hip).  --legacy times the host-side
conversion of rounds 2 - 5 instead (every parameter `.to(bf16)` on the CPU, the two LayerNorm folds as float32 mat-vecs, then uploaded).
Prints per-process seconds (checkpoint map, conversion + upload) and the wall time of the slowest."""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(path, legacy):
    t0 = time.time()
    import torch
    from classpose_amd import engine, hostinfo
    hostinfo.limit_torch_threads()
    t1 = time.time()
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda:0")
    t2 = time.time()
    sd = torch.load(path, mmap=True, weights_only=True)
    t3 = time.time()
    if legacy:
        hd = torch.bfloat16
        keep = []
        for k, v in sd.items():
            v = v.detach()
            if k.endswith(("attn.qkv.weight", "mlp.lin1.weight")):
                p = k.rsplit(".", 2)[0].rsplit(".", 1)[0]
                n = "norm1" if "qkv" in k else "norm2"
                wq, gq, btq = v.to(hd).float(), sd[f"{p}.{n}.weight"].to(hd).float(), sd[f"{p}.{n}.bias"].to(hd).float()
                wf = (wq * gq[None, :]).to(hd)
                keep += [wf.to("cuda:0"), (wq @ btq).to("cuda:0"), wf.float().sum(1).to("cuda:0")]
            else:
                keep.append(v.to(hd).contiguous().to("cuda:0"))
    else:
        w = engine.NetWeights.from_state_dict(sd, "bf16", "cuda:0")
    torch.cuda.synchronize()
    t4 = time.time()
    print("pid %d: imports %.2f s, first HIP call %.2f s, checkpoint map %.2f s, weights %.2f s" % (os.getpid(), t1 - t0, t2 - t1, t3 - t2, t4 - t3), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--legacy", action="store_true")
    ap.add_argument("--checkpoint", default="/tmp/cpx_synthetic_conic.pt")
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        child(a.checkpoint, a.legacy)
        sys.exit(0)
    if not os.path.exists(a.checkpoint):
        import torch
        from classpose_amd import synth
        os.makedirs(os.path.dirname(a.checkpoint), exist_ok=True)
        torch.save(synth.make_state_dict(7, None, depth=24, seed=0), a.checkpoint)
    t = time.time()
    ps = [subprocess.Popen([sys.executable, __file__, "--child", "--checkpoint", a.checkpoint] + (["--legacy"] if a.legacy else []))
          for _ in range(a.procs)]
    rc = [p.wait() for p in ps]
    print("%d processes, %s: wall %.2f s (incl. interpreter start + imports), exit codes %s" % (
        a.procs, "host conversion (rounds 2 - 5)" if a.legacy else "device conversion (round 6)", time.time() - t, rc), flush=True)
