"""MI355X-native BIG hot path: ctypes binding of the C-ABI HIP library + host-side packing."""
