"""TEST-ONLY torch-indexing implementation of the row-ops interface of fluent_mi355.ep.AllToAll, so that the
multi-process HOST logic (slot routing, equal-split all-to-all, expert grouping, weighted return) can run on CPU tensors
under the gloo backend.  Mirrors the contracts of csrc/ep_a2a.hip; never used by the product path."""
import torch


class TorchRowOps:
    def route_dedup(self, indices, top_k, experts_per_rank, world, cap, tok_slot, send_eid, pair_src, weights=None, send_w=None):
        send_eid.fill_(-1)
        pair_src.fill_(-1)
        if weights is not None:
            send_w.fill_(0)
        ids = indices.view(-1, top_k).tolist()
        T = len(ids)
        ts = [[-1] * world for _ in range(T)]
        pos = [0] * world
        for t in range(T):
            peers = sorted({e // experts_per_rank for e in ids[t] if 0 <= e < experts_per_rank * world})
            for d in peers:
                if pos[d] < cap:
                    ts[t][d] = d * cap + pos[d]
                pos[d] += 1
        for t in range(T):
            seen = [0] * world
            for k, e in enumerate(ids[t]):
                if 0 <= e < experts_per_rank * world:
                    d = e // experts_per_rank
                    row = ts[t][d]
                    if row >= 0:
                        send_eid[row, seen[d]] = e - d * experts_per_rank      # [rows, top_k] (possibly a view into the message tail)
                        pair_src[row * top_k + seen[d]] = t * top_k + k
                        if weights is not None:
                            send_w[row, seen[d]] = weights[t * top_k + k]
                    seen[d] += 1
        if T:
            tok_slot.copy_(torch.tensor(ts, dtype=torch.int32).view(-1))

    def sort(self, recv_eid, E, order, exclusive_sum, inverse=None):
        recv_eid = recv_eid.reshape(-1)                          # [rows, top_k] (view ok) -> the flattened (row, j) pairs
        key = torch.where((recv_eid >= 0) & (recv_eid < E), recv_eid, torch.full_like(recv_eid, E))
        order.copy_(torch.argsort(key, stable=True).to(torch.int32))
        counts = torch.bincount(key.long(), minlength=E + 1)[:E]
        exclusive_sum[0] = 0
        exclusive_sum[1:] = torch.cumsum(counts, 0).to(torch.int32)
        if inverse is not None:
            inverse[order.long()] = torch.arange(order.numel(), dtype=torch.int32)

    def gather_div(self, src, idx, n, div, dst, n_valid=None):
        if n_valid is not None:
            n = min(n, int(n_valid[0]))
        i = idx[:n].long() // div
        ok = (idx[:n] >= 0) & (i < src.shape[0])
        dst[:n][ok] = src[i[ok]]

    def send(self, x, send_slot, per_token, send_buf):
        s = send_slot.long()
        ok = (s >= 0) & (s < send_buf.shape[0])
        send_buf[s[ok]] = x[torch.arange(s.numel())[ok] // per_token]

    def combine(self, rows, slot, weights, out, per_token):
        s = slot.view(-1, per_token).long()
        w = weights.reshape(-1)[:s.numel()].view(-1, per_token)
        ok = ((s >= 0) & (s < rows.shape[0]) & (w != 0)).unsqueeze(-1)
        r = torch.where(ok, rows[s.clamp(0, max(rows.shape[0] - 1, 0))].float(), torch.zeros(())) if rows.shape[0] else torch.zeros(s.shape + (out.shape[1],))
        out.copy_((r * w.unsqueeze(-1)).sum(1).to(out.dtype))

    def gather_f32(self, vals, src, out):
        ok = (src >= 0) & (src < vals.numel())
        out.copy_(torch.where(ok, vals[src.clamp(0, max(vals.numel() - 1, 0)).long()] if vals.numel() else torch.zeros(src.numel()), torch.zeros(())).view(out.shape))
