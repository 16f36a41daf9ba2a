// dlrm.cc -- DLRM application on the FFModel shim [ref: examples/cpp/DLRM/dlrm.cc].
#include "dlrm.h"

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include "hdf5_io.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "backend.h"
#include "../../include/ffh_rng.h"

namespace {
double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
void print_vector(const std::string& name, const std::vector<int>& v) {
  std::ostringstream out;
  for (size_t i = 0; i < v.size(); i++) out << v[i] << (i + 1 < v.size() ? " " : "");
  printf("[DLRM] %s: %s\n", name.c_str(), out.str().c_str());
}
}  // namespace

DLRMConfig::DLRMConfig(void)
    : sparse_feature_size(2), sigmoid_bot(-1), sigmoid_top(-1), embedding_bag_size(1), loss_threshold(0.0f),
      arch_interaction_op("cat"), dataset_path(""), data_size(-1), optimizer("sgd"), zipf_alpha(0.0) {
  embedding_size.push_back(4);
  mlp_bot.push_back(4); mlp_bot.push_back(2);
  mlp_top.push_back(8); mlp_top.push_back(2);
}

// [ref: examples/cpp/DLRM/dlrm.cc:197-260] -- identical flags
void parse_input_args(char** argv, int argc, DLRMConfig& config) {
  auto split = [](const char* s) {
    std::vector<int> v;
    std::stringstream ss((std::string(s)));
    std::string word;
    while (std::getline(ss, word, '-')) v.push_back(std::stoi(word));
    return v;
  };
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--arch-sparse-feature-size")) { config.sparse_feature_size = atoi(argv[++i]); continue; }
    if (!strcmp(argv[i], "--arch-embedding-size")) { config.embedding_size = split(argv[++i]); continue; }
    if (!strcmp(argv[i], "--embedding-bag-size")) { config.embedding_bag_size = atoi(argv[++i]); continue; }
    if (!strcmp(argv[i], "--arch-mlp-bot")) { config.mlp_bot = split(argv[++i]); continue; }
    if (!strcmp(argv[i], "--arch-mlp-top")) { config.mlp_top = split(argv[++i]); continue; }
    if (!strcmp(argv[i], "--loss-threshold")) { config.loss_threshold = (float)atof(argv[++i]); continue; }
    if (!strcmp(argv[i], "--sigmoid-top")) { config.sigmoid_top = atoi(argv[++i]); continue; }
    if (!strcmp(argv[i], "--sigmoid-bot")) { config.sigmoid_bot = atoi(argv[++i]); continue; }
    if (!strcmp(argv[i], "--arch-interaction-op")) { config.arch_interaction_op = std::string(argv[++i]); continue; }
    if (!strcmp(argv[i], "--optimizer") && i + 1 < argc) { config.optimizer = std::string(argv[++i]); continue; }
    if (!strcmp(argv[i], "--dataset")) { config.dataset_path = std::string(argv[++i]); continue; }
    if (!strcmp(argv[i], "--data-size")) { config.data_size = atoi(argv[++i]); continue; }
    if (!strcmp(argv[i], "--zipf-alpha")) { config.zipf_alpha = atof(argv[++i]); continue; }   // not a reference flag
  }
}

// [ref: examples/cpp/DLRM/dlrm.cc:26-39]: N(0, sqrt(2/(in+out))) weights, N(0, sqrt(2/out)) bias,
// ReLU everywhere except `sigmoid_layer`
Tensor create_mlp(FFModel* model, const Tensor& input, std::vector<int> ln, int sigmoid_layer) {
  Tensor t = input;
  for (int i = 0; i < (int)(ln.size() - 1); i++) {
    float std_dev = std::sqrt(2.0f / (ln[i + 1] + ln[i]));
    Initializer* weight_init = model->own(new NormInitializer(model->next_seed(), 0, std_dev));
    std_dev = std::sqrt(2.0f / ln[i + 1]);
    Initializer* bias_init = model->own(new NormInitializer(model->next_seed(), 0, std_dev));
    ActiMode activation = i == sigmoid_layer ? AC_MODE_SIGMOID : AC_MODE_RELU;
    t = model->dense(t, ln[i + 1], activation, true /*bias*/, NULL /*weight_sharing*/, weight_init, bias_init);
  }
  return t;
}

// [ref: examples/cpp/DLRM/dlrm.cc:41-47]: U(-sqrt(1/R), sqrt(1/R)), AGGR_MODE_SUM
Tensor create_emb(FFModel* model, const Tensor& input, int input_dim, int output_dim, int idx) {
  (void)idx;
  float range = std::sqrt(1.0f / input_dim);
  Initializer* embed_init = model->own(new UniformInitializer(model->next_seed(), -range, range));
  return model->embedding(input, input_dim, output_dim, AGGR_MODE_SUM, NULL /*weight_sharing*/, embed_init);
}

// [ref: examples/cpp/DLRM/dlrm.cc:49-65]: the reference implements "cat" and asserts on "dot" (a TODO).
// "dot" here is the composition the reference's own op tests spell out for the pairwise interaction
// [ref: tests/ops/test_harness.py:125-177]: cat -> reshape [B][C][D] -> transpose -> batch_matmul -> flat,
// concatenated with the bottom-MLP output: [x | vec(Z Z^T)], width D + C*C with C = 1 + #tables.
// "dot-tril" is MLPerf-DLRM's variant: only the strict lower triangle of Z Z^T is kept (C (C - 1) / 2 products: 351 of 729
// for 26 tables; width D + C (C - 1) / 2 = 479) -- through FFModel::dot_interaction, the whole interaction as one
// MFMA kernel each way (csrc/interaction.hip); "dot-tril-ops" spells the same thing as the operator chain with
// FFModel::tril (neither operator exists in the reference).
Tensor interact_features(FFModel* model, const Tensor& x, const std::vector<Tensor>& ly, std::string interaction) {
  std::vector<Tensor> inputs;
  inputs.push_back(x);
  for (size_t i = 0; i < ly.size(); i++) inputs.push_back(ly[i]);
  if (interaction == "cat") return model->concat((int)inputs.size(), inputs.data(), 1 /*axis*/);
  if (interaction == "dot" || interaction == "dot-tril" || interaction == "dot-tril-ops") {
    const int batch = x.adim[1], d = x.adim[0], c = (int)inputs.size();
    for (const Tensor& t : inputs)
      if (t.adim[0] != d) {
        fprintf(stderr, "FATAL: --arch-interaction-op dot needs the bottom MLP output width (%d) to equal --arch-sparse-feature-size (%d)\n", d, t.adim[0]);
        abort();
      }
    Tensor cat = model->concat(c, inputs.data(), 1 /*axis*/);
    if (interaction == "dot-tril") return model->dot_interaction(cat, d);   // the chain below (with tril) as one launch each way
    Tensor z = model->reshape(cat, {batch, c, d});
    Tensor zt = model->transpose(z, {0, 2, 1});
    Tensor p = model->batch_matmul(z, zt);            // [batch][c][c]
    Tensor pf = interaction == "dot" ? model->flat(p) : model->tril(p);   // "dot-tril-ops": the operator chain, kept for parity tests
    Tensor both[2] = {x, pf};
    return model->concat(2, both, 1 /*axis*/);
  }
  fprintf(stderr, "FATAL: --arch-interaction-op %s: 'cat', 'dot', 'dot-tril' or 'dot-tril-ops'\n", interaction.c_str());
  abort();
}

// =============================================================================================
DataLoader::DataLoader(FFModel& ff, const DLRMConfig& dlrm, const std::vector<Tensor>& sparse_inputs, Tensor dense_input, Tensor label)
    : num_samples(0), next_index(0), batch_sparse_inputs(sparse_inputs), batch_dense_input(dense_input), batch_label(label),
      full_dense(nullptr), full_label(nullptr), model(&ff) {
  bag = dlrm.embedding_bag_size;
  dense_dim = dense_input.adim[0];
  full_sparse.assign(sparse_inputs.size(), nullptr);
  if (dlrm.dataset_path != "") load_hdf5(ff, dlrm);
  else generate_random(ff, dlrm);
  ff.check(ff.api->ffh_stream_sync(ff.ctx, ff.stream), "dataset sync");
}

void DataLoader::generate_random(FFModel& ff, const DLRMConfig& dlrm) {
  const bool chatty = ff.world_size <= 1 || ff.rank == 0;
  if (chatty) printf("[DLRM] Use random dataset...\n");
  if (dlrm.data_size > 0) num_samples = dlrm.data_size;
  else num_samples = 256 * 4 * std::max(1, ff.world_size) * ff.config.numNodes;   // [ref: dlrm.cc:272-276]
  const int B = ff.config.batchSize;
  if (num_samples < B) num_samples = B;
  num_samples = num_samples / B * B;
  if (chatty) printf("[DLRM] Number of random samples = %d\n", num_samples);
  const uint64_t s0 = ff.config.seed * 1000003ULL;
  const int nb = num_samples / B;
  const int64_t Bl = ff.local_batch;
  // sparse ids: owner of table t keeps ids of every sample (it gathers for the global batch)
  for (size_t t = 0; t < batch_sparse_inputs.size(); t++) {
    if (!batch_sparse_inputs[t].impl->ptr) continue;          // this rank neither owns the table nor holds a column block of it
    const int64_t n = (int64_t)num_samples * bag;
    full_sparse[t] = (int64_t*)ff.dmalloc((size_t)n * sizeof(int64_t));
    if (dlrm.zipf_alpha > 0.0) generate_zipf(ff, full_sparse[t], n, s0 + 17 + t, dlrm.embedding_size[t], dlrm.zipf_alpha);
    else ff.check(ff.api->ffh_gen_indices(ff.ctx, full_sparse[t], n, s0 + 17 + t, 0, dlrm.embedding_size[t], ff.stream), "gen_indices");
  }
  // dense features and labels: this rank's slice [rank*Bl, (rank+1)*Bl) of every batch
  full_dense = (float*)ff.dmalloc((size_t)nb * Bl * dense_dim * sizeof(float));
  full_label = (float*)ff.dmalloc((size_t)nb * Bl * sizeof(float));
  for (int k = 0; k < nb; k++) {
    const int64_t n0 = (int64_t)k * B + (int64_t)ff.rank * Bl;
    ff.check(ff.api->ffh_gen_uniform01(ff.ctx, full_dense + (int64_t)k * Bl * dense_dim, Bl * dense_dim, s0 + 5, n0 * dense_dim, ff.stream), "gen dense");
    ff.check(ff.api->ffh_gen_bernoulli(ff.ctx, full_label + (int64_t)k * Bl, Bl, s0 + 7, n0, ff.stream), "gen label");
  }
}

// --zipf-alpha A (not in the reference, whose ids are uniform: SURVEY 8d's duplicate-row stress).  Rank k of R is drawn
// with probability ~ (k+1)^-A through the inverse CDF of the continuous power law on [1, R+1); ranks are spread over the
// table by a golden-ratio multiplicative permutation so that hot rows are not neighbours.  Drawn on the host in double precision from
// the counter hash -- the same code feeds the HIP backend and the oracle backend, so both see identical ids.
void DataLoader::generate_zipf(FFModel& ff, int64_t* dst, int64_t n, uint64_t seed, int64_t rows, double alpha) {
  std::vector<int64_t> ids((size_t)n);
  const double a1 = 1.0 - alpha;
  const bool log_law = std::fabs(a1) < 1e-9;                    // alpha == 1: x = (R+1)^u
  const double top = log_law ? std::log((double)rows + 1.0) : std::pow((double)rows + 1.0, a1) - 1.0;
  uint64_t stride = (uint64_t)((double)rows * 0.6180339887498949) + 1;   // golden-ratio step, made coprime with the row count
  auto gcd = [](uint64_t a, uint64_t b) { while (b) { const uint64_t t = a % b; a = b; b = t; } return a; };
  while (gcd(stride, (uint64_t)rows) != 1) stride++;
  const uint64_t shift = ffh_hash(seed, ~0ULL) % (uint64_t)rows;
  for (int64_t i = 0; i < n; i++) {
    const double u = (double)(ffh_hash(seed, (uint64_t)i) >> 11) * (1.0 / 9007199254740992.0);
    const double x = log_law ? std::exp(u * top) : std::pow(1.0 + u * top, 1.0 / a1);
    int64_t k = (int64_t)x - 1;
    if (k < 0) k = 0;
    if (k >= rows) k = rows - 1;
    ids[(size_t)i] = (int64_t)(((uint64_t)k * (stride % (uint64_t)rows) + shift) % (uint64_t)rows);
  }
  ff.check(ff.api->ffh_memcpy_h2d(ff.ctx, dst, ids.data(), (size_t)n * sizeof(int64_t), ff.stream), "zipf ids H2D");
  ff.check(ff.api->ffh_stream_sync(ff.ctx, ff.stream), "zipf ids sync");
}

// The Criteo file of the reference [ref: examples/cpp/DLRM/dlrm.cc:279-326 (shape checks), :421-479 (H5Dread of X_cat as
// LLONG, X_int and y as FLOAT); written by preprocess_hdf.py:14-24]: X_int [N][dense] float (already log(x+1)),
// X_cat [N][tables*bag] integer, y [N] or [N][1] float.  The whole set becomes device-resident in the same layout the
// synthetic generator produces; the file is read in row chunks so that host memory stays bounded.
void DataLoader::load_hdf5(FFModel& ff, const DLRMConfig& dlrm) {
  const bool chatty = ff.world_size <= 1 || ff.rank == 0;
  if (chatty) printf("[DLRM] Start loading dataset from %s\n", dlrm.dataset_path.c_str());
  Hdf5File file(dlrm.dataset_path);
  auto bad = [&](const char* what) {
    fprintf(stderr, "FATAL: --dataset %s: %s\n", dlrm.dataset_path.c_str(), what);
    abort();
  };
  const Hdf5Dataset xi = file.describe("X_int"), xc = file.describe("X_cat"), yy = file.describe("y");
  const size_t T = batch_sparse_inputs.size();
  if (xi.dims.size() != 2 || xi.type_class != 1) bad("X_int must be a 2-D float dataset");
  if ((int)xi.dims[1] != dense_dim) bad("X_int's second dimension must equal --arch-mlp-bot[0]");                 // [ref: dlrm.cc:292]
  if (xc.dims.size() != 2 || xc.type_class != 0) bad("X_cat must be a 2-D integer dataset");
  if (xc.dims[0] != xi.dims[0] || yy.dims[0] != xi.dims[0]) bad("X_int, X_cat and y must have the same number of samples");
  if (xc.dims[1] != T * (size_t)bag) bad("X_cat's second dimension must equal (number of tables) x (bag size)");  // [ref: dlrm.cc:307]
  if (yy.dims.size() == 2 && yy.dims[1] != 1) bad("y must be [N] or [N][1]");
  const int B = ff.config.batchSize;
  uint64_t n_file = xi.dims[0];
  if (dlrm.data_size > 0 && (uint64_t)dlrm.data_size < n_file) n_file = (uint64_t)dlrm.data_size;   // --data-size caps what is loaded
  if (n_file / B == 0) bad("fewer samples than one batch");
  if (n_file / B * B > 0x7fffffffULL) bad("more than 2^31 samples");
  num_samples = (int)(n_file / B * B);       // the reference iterates num_samples / batchSize whole batches (dlrm.cc:157)
  const int nb = num_samples / B;
  const int64_t Bl = ff.local_batch;
  for (size_t t = 0; t < T; t++)
    if (batch_sparse_inputs[t].impl->ptr) full_sparse[t] = (int64_t*)ff.dmalloc((size_t)num_samples * bag * sizeof(int64_t));
  full_dense = (float*)ff.dmalloc((size_t)nb * Bl * dense_dim * sizeof(float));
  full_label = (float*)ff.dmalloc((size_t)nb * Bl * sizeof(float));

  const int batches_per_chunk = std::max(1, (1 << 18) / B);       // about 256k samples per pass over the file
  const size_t C = T * (size_t)bag;
  std::vector<int64_t> cat((size_t)batches_per_chunk * B * C), col((size_t)batches_per_chunk * B * bag);
  std::vector<float> xint((size_t)batches_per_chunk * B * dense_dim), lab((size_t)batches_per_chunk * B);
  for (int k0 = 0; k0 < nb; k0 += batches_per_chunk) {
    const int kb = std::min(batches_per_chunk, nb - k0);
    const uint64_t row0 = (uint64_t)k0 * B, rows = (uint64_t)kb * B;
    file.read_rows_i64("X_cat", row0, rows, cat.data());
    file.read_rows_f32("X_int", row0, rows, xint.data());
    file.read_rows_f32("y", row0, rows, lab.data());
    for (size_t t = 0; t < T; t++) {
      if (!full_sparse[t]) continue;
      const int64_t R = dlrm.embedding_size[t];
      for (uint64_t i = 0; i < rows; i++)
        for (int j = 0; j < bag; j++) {
          const int64_t id = cat[i * C + t * bag + j];
          // the reference gathers without a bounds check on the GPU and asserts on the CPU path (src/ops/embedding.cc:71-73)
          if (id < 0 || id >= R) {
            fprintf(stderr, "FATAL: --dataset %s: X_cat[%llu][%zu] = %lld is outside table %zu (%lld rows)\n", dlrm.dataset_path.c_str(),
                    (unsigned long long)(row0 + i), t * bag + j, (long long)id, t, (long long)R);
            abort();
          }
          col[i * bag + j] = id;
        }
      ff.check(ff.api->ffh_memcpy_h2d(ff.ctx, full_sparse[t] + row0 * bag, col.data(), rows * bag * sizeof(int64_t), ff.stream), "dataset H2D");
      ff.check(ff.api->ffh_stream_sync(ff.ctx, ff.stream), "dataset sync");     // col is reused for the next table
    }
    for (int k = 0; k < kb; k++) {
      const size_t src = (size_t)k * B + (size_t)ff.rank * Bl;                   // this rank's slice of batch k0 + k
      ff.check(ff.api->ffh_memcpy_h2d(ff.ctx, full_dense + (int64_t)(k0 + k) * Bl * dense_dim, xint.data() + src * dense_dim,
                                      (size_t)Bl * dense_dim * sizeof(float), ff.stream), "dataset H2D");
      ff.check(ff.api->ffh_memcpy_h2d(ff.ctx, full_label + (int64_t)(k0 + k) * Bl, lab.data() + src, (size_t)Bl * sizeof(float), ff.stream), "dataset H2D");
    }
    ff.check(ff.api->ffh_stream_sync(ff.ctx, ff.stream), "dataset sync");
  }
  if (chatty) {
    printf("[DLRM] Finish loading dataset from %s\n", dlrm.dataset_path.c_str());
    printf("[DLRM] Loaded %d samples\n", num_samples);
  }
}

DataLoader::~DataLoader() {
  for (int64_t* p : full_sparse) if (p) model->api->ffh_free(model->ctx, p);
  if (full_dense) model->api->ffh_free(model->ctx, full_dense);
  if (full_label) model->api->ffh_free(model->ctx, full_label);
}

// [ref: examples/cpp/DLRM/dlrm.cc:482-585, dlrm.cu:19-122]: device-to-device copies on the compute stream
void DataLoader::next_batch(FFModel& ff) {
  const int B = ff.config.batchSize;
  if (next_index + B > num_samples) next_index = 0;
  const int64_t Bl = ff.local_batch;
  const int k = next_index / B;
  ff.order_input_writes_behind_update();       // eager steps: the last step's table update may still be reading the ids
  for (size_t t = 0; t < batch_sparse_inputs.size(); t++) {
    if (!full_sparse[t]) continue;
    ff.check(ff.api->ffh_memcpy_d2d(ff.ctx, batch_sparse_inputs[t].impl->ptr, full_sparse[t] + (int64_t)next_index * bag,
                                    (size_t)B * bag * sizeof(int64_t), ff.stream), "load sparse");
  }
  ff.check(ff.api->ffh_memcpy_d2d(ff.ctx, batch_dense_input.impl->ptr, full_dense + (int64_t)k * Bl * dense_dim,
                                  (size_t)Bl * dense_dim * sizeof(float), ff.stream), "load dense");
  ff.check(ff.api->ffh_memcpy_d2d(ff.ctx, batch_label.impl->ptr, full_label + (int64_t)k * Bl, (size_t)Bl * sizeof(float), ff.stream),
           "load label");
  next_index += B;
  ff.inputs_dirty = true;
}

// =============================================================================================
// --backtrace-on-crash (debugging aid): SIGSEGV / SIGBUS / SIGABRT print the native call stack of the faulting thread before the
// process dies (glibc backtrace_symbols_fd: async-signal-safe enough for a last word; the GPU box has no core dumps to look at)
static void crash_backtrace(int sig) {
  void* frames[64];
  const int n = backtrace(frames, 64);
  const char msg[] = "\n[DLRM] fatal signal, native backtrace:\n";
  if (write(2, msg, sizeof msg - 1) < 0) {}
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

DLRMApp::DLRMApp(int argc, char** argv, const ffcomm* comm) : ff(nullptr), loader(nullptr), warmed_up(false) {
  for (int i = 1; i < argc; i++)
    if (!strcmp(argv[i], "--backtrace-on-crash")) { signal(SIGSEGV, crash_backtrace); signal(SIGBUS, crash_backtrace); signal(SIGABRT, crash_backtrace); }
  ffconfig.parse_args(argv, argc);
  if (comm) ffconfig.comm = *comm;
  parse_input_args(argv, argc, dlrm);
  const bool chatty = ffconfig.comm.world_size <= 1 || ffconfig.comm.rank == 0;
  if (chatty) {
    printf("[DLRM] batchSize(%d) workersPerNodes(%d) numNodes(%d)\n", ffconfig.batchSize, ffconfig.workersPerNode, ffconfig.numNodes);
    printf("[DLRM] EmbeddingBagSize(%d)\n", dlrm.embedding_bag_size);
    print_vector("Embedding Vocab Sizes", dlrm.embedding_size);
    print_vector("MLP Top", dlrm.mlp_top);
    print_vector("MLP Bot", dlrm.mlp_bot);
  }
  if (dlrm.embedding_size.size() > MAX_NUM_EMB) { fprintf(stderr, "FATAL: more than %d tables\n", MAX_NUM_EMB); abort(); }
  ff = new FFModel(ffconfig);

  for (size_t i = 0; i < dlrm.embedding_size.size(); i++) {
    const int dims[] = {ffconfig.batchSize, dlrm.embedding_bag_size};
    sparse_inputs.push_back(ff->create_tensor<2>(dims, DT_INT64));
  }
  {
    const int dims[] = {ffconfig.batchSize, dlrm.mlp_bot[0]};
    dense_input = ff->create_tensor<2>(dims, DT_FLOAT);
  }
  // Step 1 create dense_mlp
  Tensor x = create_mlp(ff, dense_input, dlrm.mlp_bot, dlrm.sigmoid_bot);
  std::vector<Tensor> ly;
  for (size_t i = 0; i < dlrm.embedding_size.size(); i++)
    ly.push_back(create_emb(ff, sparse_inputs[i], dlrm.embedding_size[i], dlrm.sparse_feature_size, (int)i));
  Tensor z = interact_features(ff, x, ly, dlrm.arch_interaction_op);
  create_mlp(ff, z, dlrm.mlp_top, (int)dlrm.mlp_top.size() - 2);
  if (dlrm.loss_threshold > 0.0f && dlrm.loss_threshold < 1.0f) {
    fprintf(stderr, "FATAL: --loss-threshold clamp is not implemented (the reference asserts here, dlrm.cc:125-128)\n");
    abort();
  }
  // Use SGD Optimizer [ref: examples/cpp/DLRM/dlrm.cc:130: SGDOptimizer(&ff, 0.01f), the reference driver's only choice].
  // --optimizer (this build): the other optimizers of the reference's library behind the same driver -- "sgd-momentum" =
  // SGDOptimizer(lr 0.01, momentum 0.9), "adam" = AdamOptimizer with its defaults [ref: include/optimizer.h:40-42,62-85]
  if (dlrm.optimizer == "sgd") optimizer = new SGDOptimizer(ff, 0.01f);
  else if (dlrm.optimizer == "sgd-momentum") optimizer = new SGDOptimizer(ff, 0.01f, 0.9f);
  else if (dlrm.optimizer == "adam") optimizer = new AdamOptimizer(ff);
  else { fprintf(stderr, "FATAL: --optimizer %s: 'sgd', 'sgd-momentum' or 'adam'\n", dlrm.optimizer.c_str()); abort(); }
  std::vector<MetricsType> metrics;
  metrics.push_back(METRICS_ACCURACY);
  metrics.push_back(METRICS_MEAN_SQUARED_ERROR);
  ff->compile(optimizer, LOSS_MEAN_SQUARED_ERROR_AVG_REDUCE, metrics);
  loader = new DataLoader(*ff, dlrm, sparse_inputs, dense_input, ff->label_tensor);
  ff->init_layers();
}

DLRMApp::~DLRMApp() {
  if (ff) ff->sync();
  delete loader;
  delete ff;
  delete optimizer;
}

void DLRMApp::warmup() {
  // [ref: examples/cpp/DLRM/dlrm.cc:139-149]
  loader->reset();
  ff->reset_metrics();
  loader->next_batch(*ff);
  ff->forward();
  ff->zero_gradients();
  ff->backward();
  ff->update();
  ff->sync();
  warmed_up = true;
}

void DLRMApp::train_steps(int n, bool trace) {
  for (int it = 0; it < n; it++) {
    // random input: the batch loaded in the warm-up is reused; a dataset advances every iteration, outside the trace
    // [ref: examples/cpp/DLRM/dlrm.cc:167-175]
    if (!dlrm.dataset_path.empty()) loader->next_batch(*ff);
    if (trace) ff->begin_trace(111 /*trace_id*/);
    ff->forward();
    ff->zero_gradients();
    ff->backward();
    ff->update();
    if (trace) ff->end_trace(111 /*trace_id*/);
  }
}

double DLRMApp::run_epochs() {
  if (!warmed_up) warmup();
  if (ff->config.trace_mode < 0) ff->config.trace_mode = 0;     // the driver's loop: a step is replayed only where the replay is not slower (FFConfig::trace_mode)
  const bool chatty = ff->rank == 0;
  ff->sync();   // issue_execution_fence + timing measurement
  if (ffconfig.comm.world_size > 1 && ffconfig.comm.barrier) ffconfig.comm.barrier(ffconfig.comm.user);
  if (chatty) {
    printf("[DLRM] Warmup finished...Start timer...\n");
    printf("[DLRM] Num. epochs = %d\n", ffconfig.epochs);
    printf("[DLRM] Num. iterations/epoch = %d\n", loader->num_samples / ffconfig.batchSize);
    printf("parameters.size() = %lu\n", ff->parameters.size());
  }
  const double ts_start = now_us();
  for (int epoch = 0; epoch < ffconfig.epochs; epoch++) {
    loader->reset();
    ff->reset_metrics();
    const int iterations = loader->num_samples / ffconfig.batchSize;
    train_steps(iterations, epoch > 0 /* the reference traces from the second epoch on */);
  }
  ff->sync();
  if (ffconfig.comm.world_size > 1 && ffconfig.comm.barrier) ffconfig.comm.barrier(ffconfig.comm.user);
  const double ts_end = now_us();
  const double run_time = 1e-6 * (ts_end - ts_start);
  if (chatty) {
    PerfMetrics pm = ff->get_perf_metrics();
    pm.print(ff->metrics_flags);
    // [ref: examples/cpp/DLRM/dlrm.cc:193-194] -- the reference's line as it is; a kernel library other than the product's own
    // (--backend / FFH_BACKEND_LIB: the CPU oracle in tests, an A/B build) is named on it, so a number can never be mistaken
    printf("ELAPSED TIME = %.4fs, THROUGHPUT = %.2f samples/s", run_time, loader->num_samples * (double)ffconfig.epochs / run_time);
    if (ff->api->overridden) printf("  [kernel library: %s, %s]", ff->api->ffh_backend_name(), ff->api->path.c_str());
    printf("\n");
  }
  return run_time;
}

int dlrm_main(int argc, char** argv, const ffcomm* comm) {
  DLRMApp app(argc, argv, comm);
  app.run_epochs();
  return 0;
}
