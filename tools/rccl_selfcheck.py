"""RCCL software-path rehearsal on a 1-GPU box: a world-size-1 NCCL group whose only rank sends a framed halo message to
itself with the same torch calls sphmi/slab.py uses (batch_isend_irecv of CUDA tensors, wait, device synchronise). It cannot
exercise xGMI, only that the API usage is accepted by this torch / RCCL build."""
import os
import time


def main():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dist.barrier()
    n = 170000 * 9
    frame = torch.arange(1 + n, dtype=torch.int32, device="cuda")
    frame[0] = n
    recv = torch.empty_like(frame)
    side = torch.cuda.Stream()
    for it in range(5):
        with torch.cuda.stream(side):  # the solver's kernels run on a stream of their own
            frame[1:] += 1
        side.synchronize()
        t0 = time.perf_counter()
        ops = [dist.P2POp(dist.isend, frame[:1 + n], 0), dist.P2POp(dist.irecv, recv[:1 + n], 0)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert int(recv[0].item()) == n and torch.equal(recv, frame), "self send/recv mismatch"
        print("iteration %d: %.1f us for %.1f MB" % (it, dt * 1e6, 4 * (1 + n) / 1e6), flush=True)
    dist.destroy_process_group()
    print("rccl self test ok")


if __name__ == "__main__":
    main()
