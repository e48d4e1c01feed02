"""`flashinfer.comm.all_gather` (flashinfer_comm_fusion.py:50-58, 271-283)."""
from fluent_mi355.comm import create_ipc_workspace_for_allgather, simple_all_gather  # noqa: F401
