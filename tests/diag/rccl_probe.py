"""Probe: does RCCL ("nccl" backend) accept the ranks this repo can give it on a one-GPU box?  world 1, then world 2 on the
same device.  python tests/diag/rccl_probe.py <world>"""
import os, sys, socket
import torch, torch.distributed as dist, torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from vllm_ltr_amd.distributed import gather_scores
    counts = [5 + 3 * r for r in range(world)]
    local = torch.arange(counts[rank], dtype=torch.float32, device=dev) + 100 * rank
    out = torch.empty(sum(counts), dtype=torch.float32, device=dev)
    g = gather_scores(local, counts, out=out)
    # the form ShardedScorer uses when a collective timeout is set: async op + a host-side wait with a deadline
    import datetime
    g2 = gather_scores(local, counts, wait=lambda w: w.wait(datetime.timedelta(seconds=30)))
    assert torch.equal(g, g2)
    t = torch.tensor([rank], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    f = torch.tensor([1.5 + rank], dtype=torch.float64, device=dev)
    dist.all_reduce(f, op=dist.ReduceOp.MAX)
    b = torch.full((7,), float(rank + 1), device=dev)
    dist.broadcast(b, src=0)
    dist.barrier()
    torch.cuda.synchronize()
    print(f"rank {rank}/{world}: gather {g.tolist()} max-rank {int(t)} max-f64 {float(f)} bcast {b[0].item()}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.start_processes(worker, args=(world, port), nprocs=world, start_method="spawn")
    print("ok world", world)
