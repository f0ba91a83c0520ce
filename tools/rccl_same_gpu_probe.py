"""Can two ranks share ONE GPU under the nccl (= RCCL) backend?  (NCCL proper refuses: 'Duplicate GPU detected'.)"""
import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)
    try:
        dist.init_process_group('nccl', rank=rank, world_size=world)
        t = torch.full((1024,), float(rank + 1), device='cuda:0')
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print('rank', rank, 'all_reduce ->', float(t[0]), flush=True)
        dist.destroy_process_group()
    except Exception as e:
        print('rank', rank, 'FAIL', type(e).__name__, str(e)[:400].replace('\n', ' | '), flush=True)
if __name__ == '__main__':
    mp.spawn(w, args=(2, 29533), nprocs=2, join=True)
