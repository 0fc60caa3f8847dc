import sys, torch, pytest
torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
sys.exit(pytest.main(["tests/test_hip_unetpp.py", "-q", "-x", "-k", "test_conv_bn_node_padded"]))
