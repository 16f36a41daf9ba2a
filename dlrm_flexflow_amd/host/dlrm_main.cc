// dlrm_main.cc -- `dlrm` executable: examples/cpp/DLRM's top_level_task as a plain main().
// Single-GPU only (one process = one GPU); multi-GPU runs go through run_dlrm.py / bench.py,
// which supply the RCCL collectives.  [ref: src/runtime/cpp_driver.cc:22-44, examples/cpp/DLRM/dlrm.cc:77-195]
#include "dlrm.h"
int main(int argc, char** argv) { return dlrm_main(argc, argv, nullptr); }
