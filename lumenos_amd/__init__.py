"""lumenos_amd: MI355X-native server-side homomorphic Ligero prover (hot path of ChainSafe/lumenos).

The product is the C-ABI HIP library in csrc/ (include/lumenos_hip.h); this
package only carries its build script, a ctypes binding and the host-side
parameter logic used by the Python harnesses.
"""
__all__ = ["hip", "params"]
