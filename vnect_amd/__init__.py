"""vnect_amd: MI355X-native VNect inference path behind the reference's VNectEstimator call surface."""
from .estimator import VNectEstimator  # noqa: F401  (mirrors /root/reference/__init__.py:1)
