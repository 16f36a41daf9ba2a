// dlrm_main.cc -- `dlrm` executable: examples/cpp/DLRM's top_level_task as a plain main().
// `-ll:gpu N` with N > 1 starts N ranks of this binary, one per GPU, with the RCCL collectives set up from C++
// (launcher.cc); bench.py / run_dlrm.py are the Python launchers of the same application.
// [ref: src/runtime/cpp_driver.cc:22-44, examples/cpp/DLRM/dlrm.cc:77-195, examples/cpp/DLRM/run_random.sh:3]
#include "launcher.h"
int main(int argc, char** argv) { return dlrm_launch(argc, argv); }
