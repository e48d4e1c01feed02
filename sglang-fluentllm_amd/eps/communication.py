"""eps.communication — TPDPConvertor (call site python/sglang/srt/layers/dp_attention.py:62-74) and the process-wide communicator
the reference builds at start-up (srt/distributed/parallel_state.py:52,963-977)."""
from fluent_mi355.comm import MscclppCommunicator, MscclppCommunicatorParams, TPDPConvertor  # noqa: F401
