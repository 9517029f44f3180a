"""signal_amd: MI355X-native (gfx950) implementation of the Signal multi-modal ReID hot path.

Host code mirrors the reference's Python interface (make_frame / Signal.forward / engine.processor);
the compute is hand-written HIP behind the C ABI in include/signal_hip.h."""
__version__ = "0.1.0"
