#!/usr/bin/env python3
"""Can two ranks of one RCCL communicator share ONE GPU on this box?  (The pool gives one GPU per call: if they can, the multi-rank
RCCL paths -- the gather of bench.py, the all-to-all of the sharded tile -- get their first run with a real peer; if RCCL refuses
duplicate devices, the answer is recorded and the paths stay world-1-only on hardware.)   python tools/rccl_same_device_probe.py"""
import os
import subprocess
import sys

WORKER = r"""
import os, sys, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
x = torch.full((1024,), float(rank + 1), device="cuda")
dist.all_reduce(x)
torch.cuda.synchronize()
send = torch.arange(world * 4, dtype=torch.float32, device="cuda") + 100 * rank
recv = torch.empty_like(send)
dist.all_to_all_single(recv, send)
outs = [torch.empty(8, device="cuda") for _ in range(world)] if rank == 0 else None
dist.gather(torch.full((8,), float(rank), device="cuda"), outs, dst=0)
torch.cuda.synchronize()
print("RANK", rank, "allreduce", float(x[0]), "a2a", recv.tolist(), "gather", [float(o[0]) for o in outs] if outs else None, flush=True)
dist.destroy_process_group()
"""


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29631",
                   NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    rc = 0
    for r, p in enumerate(procs):
        try:
            out, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += "\n[timeout]"
        print(f"--- rank {r} rc={p.returncode}\n{out[-3000:]}")
        rc |= (p.returncode or 0) != 0
    print("SAME_DEVICE_RCCL", "REFUSED_OR_FAILED" if rc else "WORKS")


if __name__ == "__main__":
    main()
