// Layout and flag arithmetic of the one-shot peer-mapped exchange (comm_oneshot.hip) — shared by the device kernel and
// the host emulation of the same protocol (fl_comm_host_exchange: what the CPU-side two-process protocol test drives).
//
// Every rank owns ONE workspace that its peers map (hipIpc) and WRITE into:
//   [state 256 B][flags u32 [2 parity][world + 1][max(max_tokens, world)]][inbox bf16 [2 parity][world][max_tokens][hidden]]
// (a flag row has max(max_tokens, world) entries: the sync row is indexed by source RANK, so with max_tokens < world a
//  row of max_tokens entries would alias the neighbouring rows)
// An operation with epoch e (1, 2, 3, ... — the same on every rank: all ranks issue the same sequence) uses parity e & 1:
//   push : source rank s writes its row for destination row r into the DESTINATION's inbox[parity][s][r], fences
//          (system scope), then stores e into the destination's flags[parity][s][r];
//   sync : every rank also stores e into every peer's flags[parity][world][s] (row index = source rank) and waits for
//          all peers' — a rank that has nothing to receive in this operation (reduce-scatter with fewer tokens than
//          ranks) must still not run ahead: finishing operation e then implies that every peer has STARTED e, i.e. has
//          finished consuming e-1;
//   wait : the consumer of row r spins until flags[parity][s][r] == e for every s, then reduces inbox[parity][0..W)[r]
//          in rank order (the same order on every rank: all ranks get bit-identical sums).
// Why two parities are enough: a rank starts pushing e+2 (same parity as e) only after it finished e+1, which needed
// every peer's e+1 sync flag, which a peer stores only after its own operation e has completed (stream order).
// The epoch lives in the workspace (state.epoch) and is advanced by the last workgroup of the kernel itself, so that a
// captured hipGraph replays correctly (no host-side counter baked into kernel arguments).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FL_HD __host__ __device__ inline
#else
#define FL_HD static inline
#endif

struct FlCommState {       // first 256 bytes of a workspace
  uint32_t epoch;          // epoch of the NEXT operation (starts at 1)
  uint32_t arrive;         // workgroups of the running kernel that are done
  uint32_t error;          // != 0: a wait timed out.  STICKY and fatal for this communicator: the failing launch poisons its
                           // outputs with NaN, does not advance the epoch, and every later launch poisons its outputs at
                           // once without touching a peer (the ranks are out of step; re-create the communicator)
  uint32_t pad[61];
};

struct FlCommLayout {
  int world;
  long long max_tokens;
  int hidden;
};

FL_HD long long fl_comm_flags_offset() { return 256; }
FL_HD long long fl_comm_flag_cols(const FlCommLayout L) { return L.max_tokens > L.world ? L.max_tokens : (long long)L.world; }
FL_HD long long fl_comm_flag_index(const FlCommLayout L, int parity, int src /*0..world; world = the sync row*/, long long row) {
  return ((long long)parity * (L.world + 1) + src) * fl_comm_flag_cols(L) + row;
}
FL_HD long long fl_comm_inbox_offset(const FlCommLayout L) {
  const long long flags = 2ll * (L.world + 1) * fl_comm_flag_cols(L) * 4;
  return 256 + ((flags + 255) / 256) * 256;
}
FL_HD long long fl_comm_inbox_row(const FlCommLayout L, int parity, int src, long long row) {   // in bf16 elements from the inbox base
  return (((long long)parity * L.world + src) * L.max_tokens + row) * L.hidden;
}
FL_HD long long fl_comm_workspace_bytes(const FlCommLayout L) {
  return fl_comm_inbox_offset(L) + 2ll * L.world * L.max_tokens * L.hidden * 2;
}
// token slices of a reduce-scatter = get_num_tokens_per_rank (flashinfer_comm_fusion.py:237-244): the first T % W ranks own
// one token more
FL_HD long long fl_comm_slice_lo(long long T, int world, int r) {
  const long long base = T / world, rem = T % world;
  return r * base + (r < rem ? r : rem);
}
FL_HD int fl_comm_owner(long long T, int world, long long t) {
  const long long base = T / world, rem = T % world;
  const long long big = rem * (base + 1);
  if (t < big) return (int)(t / (base + 1));
  return (int)(rem + (base ? (t - big) / base : 0));
}
