// launcher.h -- `dlrm -ll:gpu N` as N processes, one per GPU, started by the binary itself (launcher.cc).
#pragma once
// Runs the DLRM driver: in this process for -ll:gpu <= 1, else as N child ranks with an RCCL communicator built from
// C++.  Returns the process exit code.
int dlrm_launch(int argc, char** argv);
