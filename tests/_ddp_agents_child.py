"""Child process of tests/test_gpu_ddp_agents.py: ONE rank of a data-parallel job that trains the real DetectorAgent and
DenoiserAgent together (agent.train_concurrent: two models, two HIP streams, their gradient collectives interleaved on
one communicator).  Launched with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment; backend nccl (RCCL)
when every rank has its own GPU, else gloo on the GPU tensors of the one shared GPU.  Writes <out>/rank<r>.pt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    out_dir, precision = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    own_gpu = torch.cuda.device_count() >= world
    torch.cuda.set_device(rank if own_gpu else 0)
    import sos_amd
    from sos_amd import agent
    from sos_amd.common import MyConfig
    from sos_amd.dataset import make_batch
    from sos_amd.denoiser import networks as jnet
    from sos_amd.detector import networks as dnet
    sos_amd.set_precision(precision)
    B = 2
    bd = make_batch("detector", 9000 + 100 * rank, B)            # every rank its own clips
    bj = make_batch("denoiser", 9500 + 100 * rank, B)

    def nets(seed):
        torch.manual_seed(seed)
        return dnet.get_network().cuda(), jnet.get_network(MyConfig()).cuda()

    # ---- the rank's stand-alone gradients from rank 0's initial weights (no process group yet: plain agents)
    det0, jm0 = nets(0)
    local = {}
    for tag, cls, net, batch in (("det", agent.DetectorAgent, det0, bd), ("jm", agent.DenoiserAgent, jm0, bj)):
        ag = cls(net.train(), lr=1e-3)
        assert ag.bucketer is None
        _, losses = ag.forward(batch)
        ag.optimizer.zero_grad(set_to_none=True)
        sum(losses.values()).backward()
        local[tag] = {n: p.grad.detach().clone().cpu() for n, p in net.named_parameters()}
    del det0, jm0

    # ---- the data-parallel job: replicas start from DIFFERENT weights, rank 0's win (broadcast_module_state)
    dist.init_process_group("nccl" if own_gpu else "gloo", rank=rank, world_size=world,
                            **({"device_id": torch.device("cuda", rank)} if own_gpu else {}))
    det, jm = nets(rank)
    ag_det = agent.DetectorAgent(det.train(), lr=1e-3)
    ag_jm = agent.DenoiserAgent(jm.train(), lr=1e-3)
    assert ag_det.bucketer is not None and ag_jm.bucketer is not None and ag_det.world == world
    assert ag_jm.bucketer.collective and abs(ag_jm.optimizer.grad_scale - 1.0 / world) < 1e-12
    grads1 = None
    for step in range(2):
        agent.train_concurrent([(ag_jm, bj), (ag_det, bd)])
        if step == 0:            # p.grad = the all-reduced SUM in the bucket views (the 1/world lives in the Adam kernel)
            grads1 = {"det": {n: p.grad.detach().clone().cpu() for n, p in det.named_parameters()},
                      "jm": {n: p.grad.detach().clone().cpu() for n, p in jm.named_parameters()}}
    torch.cuda.synchronize()
    state = {"det": {k: v.detach().cpu() for k, v in det.state_dict().items()},
             "jm": {k: v.detach().cpu() for k, v in jm.state_dict().items()}}
    torch.save({"rank": rank, "world": world, "backend": dist.get_backend(), "local": local, "ddp_sum": grads1, "state": state,
                "param_names": {"det": [n for n, _ in det.named_parameters()], "jm": [n for n, _ in jm.named_parameters()]}},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
