"""torch.distributed over gloo for tensors that live on ONE shared GPU: every collective is staged through
the host.  Test infrastructure: it lets DistHotPath run its real HIP entry points (byte tables, overflow lists,
sp_table_merge, slot-range filter, shared page-locked segment, device window table) with 2-3 processes on a
single-GPU box, where RCCL refuses two ranks on one device."""


class _Done:
    def wait(self):
        return True


class StagedDist:
    def __init__(self, dist):
        self.d = dist

    def get_rank(self):
        return self.d.get_rank()

    def get_world_size(self):
        return self.d.get_world_size()

    def barrier(self):
        self.d.barrier()

    def broadcast_object_list(self, objs, src=0):
        self.d.broadcast_object_list(objs, src=src)

    def all_reduce(self, ten):
        c = ten.cpu()
        self.d.all_reduce(c)
        ten.copy_(c)

    def broadcast(self, ten, src=0):
        c = ten.cpu()
        self.d.broadcast(c, src=src)
        ten.copy_(c)

    def all_gather(self, outs, ten):
        co = [o.cpu() for o in outs]
        self.d.all_gather(co, ten.cpu().contiguous())
        for o, c in zip(outs, co):
            o.copy_(c)

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None, async_op=False):
        co = out.cpu().contiguous()
        self.d.all_to_all_single(co, inp.cpu().contiguous(), output_split_sizes, input_split_sizes)
        out.copy_(co)
        return _Done() if async_op else None
