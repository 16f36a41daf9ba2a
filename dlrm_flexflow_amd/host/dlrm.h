// dlrm.h -- the examples/cpp/DLRM driver re-stated on the FFModel shim: same flags, same model
// topology, same warm-up + timed-epoch protocol, same "ELAPSED TIME ... THROUGHPUT" line.
// [ref: examples/cpp/DLRM/dlrm.h:24-89, examples/cpp/DLRM/dlrm.cc]
#pragma once

#include <string>
#include <vector>

#include "ffmodel.h"

#define MAX_NUM_EMB 1000

struct DLRMConfig {
  DLRMConfig(void);   // defaults of [ref: examples/cpp/DLRM/dlrm.h:25-36]
  int sparse_feature_size, sigmoid_bot, sigmoid_top, embedding_bag_size;
  float loss_threshold;
  std::vector<int> embedding_size, mlp_bot, mlp_top;
  std::string arch_interaction_op, dataset_path;
  int data_size;
  std::string optimizer;   // --optimizer sgd (default, the reference driver's) | sgd-momentum | adam (not a reference flag)
  double zipf_alpha;   // > 0: synthetic ids follow a power law instead of the reference's uniform draw (not a reference flag)
};

void parse_input_args(char** argv, int argc, DLRMConfig& config);

Tensor create_mlp(FFModel* model, const Tensor& input, std::vector<int> ln, int sigmoid_layer);
Tensor create_emb(FFModel* model, const Tensor& input, int input_dim, int output_dim, int idx);
Tensor interact_features(FFModel* model, const Tensor& x, const std::vector<Tensor>& ly, std::string interaction);

// Dataset (synthetic, or the reference's HDF5 Criteo file with --dataset) resident on the device (the reference keeps it in zero-copy host memory and
// gathers + copies H2D every batch, [ref: examples/cpp/DLRM/dlrm.cc:357-377, dlrm.cu:19-122]).
// Distributions of [ref: examples/cpp/DLRM/dlrm.cc:413-420] from the seeded counter RNG, with a
// per-table row count (the reference asserts all tables equal, :349-354 -- lifted).
class DataLoader {
 public:
  DataLoader(FFModel& ff, const DLRMConfig& dlrm, const std::vector<Tensor>& sparse_inputs, Tensor dense_input, Tensor label);
  ~DataLoader();
  void next_batch(FFModel& ff);
  void shuffle() {}
  void reset() { next_index = 0; }
  int num_samples, next_index;

 private:
  void generate_random(FFModel& ff, const DLRMConfig& dlrm);
  void generate_zipf(FFModel& ff, int64_t* dst, int64_t n, uint64_t seed, int64_t rows, double alpha);
  void load_hdf5(FFModel& ff, const DLRMConfig& dlrm);      // --dataset: X_int / X_cat / y of the reference's Criteo file
  std::vector<Tensor> batch_sparse_inputs;
  Tensor batch_dense_input, batch_label;
  std::vector<int64_t*> full_sparse;   // per owned table: [num_samples][bag]
  float *full_dense, *full_label;      // this rank's samples of every batch: [num_samples/world][...]
  int bag, dense_dim;
  FFModel* model;
};

// The whole application: what top_level_task builds [ref: examples/cpp/DLRM/dlrm.cc:77-195].
struct DLRMApp {
  FFConfig ffconfig;
  DLRMConfig dlrm;
  FFModel* ff;
  DataLoader* loader;
  Optimizer* optimizer = nullptr;
  std::vector<Tensor> sparse_inputs;
  Tensor dense_input;
  bool warmed_up;
  DLRMApp(int argc, char** argv, const ffcomm* comm);
  ~DLRMApp();
  void warmup();                 // the reference's single warm-up iteration
  void train_steps(int n, bool trace);   // n x {forward, zero_gradients, backward, update}
  double run_epochs();           // the timed loop; returns elapsed seconds and prints the THROUGHPUT line
};

int dlrm_main(int argc, char** argv, const ffcomm* comm);
