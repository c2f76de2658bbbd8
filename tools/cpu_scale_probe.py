import os, sys, time, json
ROOT=os.getcwd()
sys.path.insert(0, ROOT)
import bench
import multiprocessing as mp
def run(P, threads, per, pin=True, numa=False):
    ctx = mp.get_context('spawn'); barrier, out = ctx.Barrier(P), ctx.Queue(); ps=[]
    avail = sorted(os.sched_getaffinity(0))
    for r in range(P):
        cores = set(avail[r*threads:(r+1)*threads]) if pin else None
        ps.append(ctx.Process(target=bench._cpu_shard_worker, args=(r, cores, threads, dict(num_classes=14, height=64, width=128), per, barrier, out))); ps[-1].start()
    ts=[out.get(timeout=900)[1] for _ in range(P)]
    for p in ps: p.join()
    return ts
if __name__ == '__main__':
    print('cpus', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), flush=True)
    os.system("lscpu | egrep 'Model name|Socket|NUMA|Thread|Core' ")
    for P, th, per in [(1,16,8),(2,16,8),(4,16,8),(8,16,8),(8,16,64),(16,8,8),(4,32,16)]:
        t0=time.time(); ts=run(P, th, per)
        print(json.dumps(dict(P=P, threads=th, per=per, slowest=max(ts) if all(isinstance(t,float) for t in ts) else ts, ips=P*per/max(ts) if all(isinstance(t,float) for t in ts) else None, wall=time.time()-t0)), flush=True)
