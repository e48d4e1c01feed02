// fl_mla_decode — C-ABI dispatch over the KV-cache formats of MLATokenToKVPool (memory_pool.py:635-658).
#include "fl_common.h"

int fl_mla_decode_fp8_impl(const FlMlaDecodeArgs* a, hipStream_t stream);
int fl_mla_decode_bf16_impl(const FlMlaDecodeArgs* a, hipStream_t stream);   // mla_decode_bf16.hip

extern "C" int fl_mla_decode(const FlMlaDecodeArgs* args, fl_stream_t stream) {
  FL_CHECK_ARG(args != nullptr, "fl_mla_decode: null args");
  switch (args->kv_format) {
    case FL_KV_FP8_PER_TOKEN:
    case FL_KV_FP8_576:
      return fl_mla_decode_fp8_impl(args, (hipStream_t)stream);
    case FL_KV_BF16_576:
      return fl_mla_decode_bf16_impl(args, (hipStream_t)stream);
    default:
      fl_set_error("fl_mla_decode: kv_format %d not implemented", args->kv_format);
      return FL_ERR_UNSUPPORTED;
  }
}
