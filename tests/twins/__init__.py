"""Python mirrors of the C++ hosts and command lines (csrc/svr_host.cpp, pvr_host.cpp, svr_cli.cpp, pvr_cli.cpp, svr_slic.h, irtk_reg.cpp's
geometry): TEST SCAFFOLDING.  They drive either the HIP engine or the oracle through the same operator surface, so that a parity test can
run "the same loop on both sides", and they carry the CPU (gloo) tests of the sharded paths.  Nothing in fetalreconstruction_amd/ imports
them; the product's hosts are the C++ ones."""
