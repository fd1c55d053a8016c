"""float64 torch twin of one dense assembly pass -- moved to oracle/torch_port.py (bench.py's CPU baseline shares it);
this module keeps the import path the tests use."""
from oracle.torch_port import dense_assemble, grad_fixed  # noqa: F401
