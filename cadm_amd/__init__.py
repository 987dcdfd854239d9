"""cadm_amd: MI355X-native implementation of CaDM's CEM-planning / ensemble-training hot path.

Importing the package does not load the HIP library; constructing a model does, and fails
loudly when libcadm_hip.so is not built or no GPU is visible (there is no CPU fallback).
"""
__version__ = "0.1.0"
