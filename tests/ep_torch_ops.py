"""TEST-ONLY torch-indexing implementation of the row-ops interface of fluent_mi355.ep.AllToAll, so that the
multi-process HOST logic (slot routing, equal-split all-to-all, expert grouping, weighted return) can run on CPU tensors
under the gloo backend.  Mirrors the contracts of csrc/ep_a2a.hip; never used by the product path."""
import torch


class TorchRowOps:
    def route(self, indices, experts_per_rank, world, cap, send_slot, send_eid):
        send_eid.fill_(-1)
        cursor = [0] * world
        for p, e in enumerate(indices.tolist()):
            slot = -1
            if 0 <= e < experts_per_rank * world:
                d = e // experts_per_rank
                if cursor[d] < cap:
                    slot = d * cap + cursor[d]
                    send_eid[slot] = e - d * experts_per_rank
                    cursor[d] += 1
            send_slot[p] = slot

    def sort(self, recv_eid, E, order, exclusive_sum):
        key = torch.where((recv_eid >= 0) & (recv_eid < E), recv_eid, torch.full_like(recv_eid, E))
        order.copy_(torch.argsort(key, stable=True).to(torch.int32))
        counts = torch.bincount(key.long(), minlength=E + 1)[:E]
        exclusive_sum[0] = 0
        exclusive_sum[1:] = torch.cumsum(counts, 0).to(torch.int32)

    def gather(self, src, idx, n, dst):
        i = idx[:n].long()
        ok = (i >= 0) & (i < src.shape[0])
        dst[:n][ok] = src[i[ok]]

    def scatter(self, src, idx, n, dst):
        i = idx[:n].long()
        ok = (i >= 0) & (i < dst.shape[0])
        dst[i[ok]] = src[:n][ok]

    def send(self, x, send_slot, top_k, send_buf):
        s = send_slot.long()
        ok = (s >= 0) & (s < send_buf.shape[0])
        send_buf[s[ok]] = x[torch.arange(s.numel())[ok] // top_k]

    def combine(self, ret, send_slot, weights, out, top_k):
        s = send_slot.view(-1, top_k).long()
        ok = (s >= 0).unsqueeze(-1)
        rows = ret[s.clamp_min(0)].float() * ok
        out.copy_((rows * weights.view(-1, top_k, 1)).sum(1).to(out.dtype))
