"""hipdp -- host-side binding of libdpp_hip.so (the gfx950 kernels) for the net/ and trainer/ packages."""
