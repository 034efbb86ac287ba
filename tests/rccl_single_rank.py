"""Child process of test_gpu_train.py::test_rccl_calls_of_the_data_parallel_helpers_on_one_rank: the collectives the
data-parallel helpers issue (flat f32 all-reduce of the gradient bucket, parameter / projection / counter broadcasts,
the f64 MAX of the step time, the int32 MAX of the bad-token flag) through RCCL on device tensors, with a process
group of ONE rank -- the only RCCL configuration a one-GPU box can run.  The helpers skip their collectives at world
size 1, so the world size they see is patched to 2: sums and broadcasts over one rank are identities, which is what
the checks below rely on.  Prints 'ok' at the end."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method="env://", world_size=1, rank=0)
    try:
        from scmomer_amd import dist as sdist
        import scmomer_amd as S
        real_world = dist.get_world_size
        dist.get_world_size = lambda group=None: 2            # make the helpers issue their collectives
        model = S.PerformerLM(num_tokens=7, dim=200, depth=1, max_seq_len=64, heads=10, local_attn_heads=0,
                              g2v_position_emb=False).cuda()
        params = [p for p in model.parameters() if p.requires_grad]
        before = [p.detach().clone() for p in params]
        sdist.broadcast_parameters(list(model.parameters()) + list(model.buffers()), src=0)
        assert all(torch.equal(a, b) for a, b in zip(before, params))
        avg = sdist.GradientAverager(params)
        assert avg.world == 2 and avg.flat.is_cuda
        for v in avg.views:
            v.fill_(3.0)
        for p, v in zip(params, avg.views):
            p.grad = v
        avg.average_()                                        # sum over the one rank, divided by the patched 2
        assert float(avg.flat.min()) == 1.5 and float(avg.flat.max()) == 1.5
        avg.sum_in_place_()
        assert float(avg.flat.min()) == 1.5
        perf = model.performer
        mats = [m.projection_matrix.clone() for m in perf.modules() if hasattr(m, "projection_matrix")]
        perf.calls_since_last_redraw.fill_(7)
        sdist.broadcast_projections(perf, src=0)
        assert int(perf.calls_since_last_redraw) == 7
        assert all(torch.equal(a, m.projection_matrix) for a, m in
                   zip(mats, [m for m in perf.modules() if hasattr(m, "projection_matrix")]))
        assert sdist.max_over_ranks(0.25, torch.device("cuda", 0)) == 0.25
        flag = torch.tensor([1], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        assert int(flag) == 1
        dist.barrier()
        torch.cuda.synchronize()
        dist.get_world_size = real_world
        print("ok")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
