"""Host enqueue time vs wall time of one training step, plain and with the data-parallel schedule on a one-rank
RCCL communicator (EVC_DP_FORCE=1): what the collectives' host-side calls cost when no byte has to move.

    python scripts/dp_host_probe.py          # both modes, one after the other (two processes)
"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    from bench import synthetic_inputs
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    dev = "cuda:0"
    torch.cuda.set_device(0)
    if os.environ.get("EVC_DP_FORCE") == "1":
        torch.distributed.init_process_group("nccl", device_id=torch.device(dev))
    g = DistillGraph(256, every_n=10, device=dev)
    batches = [synthetic_inputs(256, 300, 1152, 4716, 100 + i, dev, False) for i in range(4)]
    nhost = [b[1].cpu().numpy() for b in batches]
    for i in range(4):
        x, n, y = batches[i % 4]
        g.step(x, y, n, num_frames_host=nhost[i % 4])
    torch.cuda.synchronize()
    K = 20
    host = 0.0
    t0 = time.perf_counter()
    for i in range(K):
        x, n, y = batches[i % 4]
        h0 = time.perf_counter()
        g.step(x, y, n, num_frames_host=nhost[i % 4])
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print("dp=%s: host enqueue %.2f ms/step, wall %.2f ms/step" % (g.dp, host / K * 1e3, wall / K * 1e3), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for force in ("0", "1"):
            env = dict(os.environ, EVC_DP_FORCE=force, MASTER_ADDR="127.0.0.1", MASTER_PORT="29519", RANK="0", WORLD_SIZE="1")
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
