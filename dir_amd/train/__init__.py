"""Training-side host code: the backward pass over the joint-token path, composed from libdir_hip.so kernels."""
