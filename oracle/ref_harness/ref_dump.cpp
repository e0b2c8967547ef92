// ref_dump.cpp -- the route to a PINNED oracle: runs the REFERENCE's own tloam::LocalRegistration
// (/root/reference/src/models/registration/registration.cpp:879-1133, compiled from where it lies, unmodified) on the
// inputs of the committed golden cases and dumps what the real implementation produces, so that
// tests/test_golden_ref.py can hold the C oracle, the numpy restatement and the HIP path against it.
//
// TEST INFRASTRUCTURE, and NOT buildable in the round's image: the reference needs Eigen3, Ceres 2.0, Open3D 0.12,
// yaml-cpp and the ROS logging macros (registration.hpp:17-31), none of which exist here; the CMakeLists.txt beside
// this file finds the real packages and fails at configure time without them (no stand-ins).  On a machine that has
// them:   cmake -S oracle/ref_harness -B oracle/_ref/build -DTLOAM_REFERENCE_DIR=/path/to/tloam && cmake --build oracle/_ref/build
//         (or, with docker and a network, ONE command: oracle/ref_harness/run.sh -- see the Dockerfile beside this file)
//         python tests/golden_ref_tools/export_ref_inputs.py          # tests/golden/case_*.npz -> oracle/_ref/in/*.bin
//         oracle/_ref/build/ref_dump oracle/_ref/in tests/golden_ref  # -> tests/golden_ref/case_*.ref.txt
//         python -m pytest tests/test_golden_ref.py
//
// What the reference lets an outside caller observe is the result pose only (the weights and residual slots are locals
// of scanMatching, the correspondence lists live inside the ceres::Problem).  The per-outer-iteration chain is
// therefore recovered by running the SAME inputs with `max_iterations` = 1, 2, ... n_outer: the pose after k outer
// iterations is `T_result` of the k-iteration run (the GNC schedule of iteration k does not depend on max_iterations).
// Cases whose predicted rotation is below 1e-2 rad are skipped: the reference perturbs them with
// Eigen::Vector3d::Random() (registration.cpp:884-886), which the caller cannot control.
//
// Input file (little endian), written by tests/golden_ref_tools/export_ref_inputs.py:
//   int32 n_outer; double cfg[16] (the TLS: keys in the order of tloam_tls_config); double T_pred[16] (row-major);
//   then 8 clouds in the order src planar, ground, edge, sphere, tgt planar, ground, edge, sphere: int64 n; double xyz[3n].
//
// The minimiser's bookkeeping -- where parity is actually won or lost -- is dumped as well, again
// without touching the reference: scanMatching keeps its ceres::Solver::Summary to itself (registration.cpp:1046-1047), so
// this file DEFINES ceres::Solve(options, problem, summary).  The reference's object file binds to that definition (an
// executable's own symbols come first), which forwards to the real one in libceres (dlsym RTLD_NEXT on the mangled name)
// and then writes the Summary: one "s" line per Solve call, one "i" line per minimiser iteration (cost, cost change, step
// accepted or rejected, trust-region radius, step norm, step quality, gradient max norm), and -- through an IterationCallback the
// interposer adds to a COPY of the options -- one "x" line per iteration with the state the minimiser is at (StateDump below).
// libceres must be linked SHARED for this (the CMake recipe checks).
#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <utility>
#include <vector>

#include <yaml-cpp/yaml.h>

#include "tloam/models/registration/registration.hpp"

#include <ceres/ceres.h>

namespace {
std::FILE* g_solve_log = nullptr;   // where the interposed ceres::Solve writes (the case's .ref.txt)
int g_run = 0, g_call = 0;          // max_iterations of the current run, index of the Solve call inside it
}  // namespace

namespace {
// The state the minimiser is at after every iteration (the accepted iterate: a rejected candidate leaves it unchanged), one
// "x" line per iteration: run, Solve call, iteration, the parameter block(s), step quality and gradient max norm.  With it the
// reject-four-times dynamics of SURVEY Appendix A.13 are compared STATE BY STATE, not only through the Summary's counters.
// Needs Solver::Options::update_state_every_iteration (a copy-out after each iteration: no effect on the iterates) -- set on
// a COPY of the reference's options inside the interposed Solve; the reference's source stays untouched.
class StateDump : public ceres::IterationCallback {
 public:
  StateDump(std::vector<double*> blocks, std::vector<int> sizes) : blocks_(std::move(blocks)), sizes_(std::move(sizes)) {}
  ceres::CallbackReturnType operator()(const ceres::IterationSummary& s) override {
    if (g_solve_log) {
      std::fprintf(g_solve_log, "x %d %d %d", g_run, g_call, s.iteration);
      for (size_t b = 0; b < blocks_.size(); ++b)
        for (int i = 0; i < sizes_[b]; ++i) std::fprintf(g_solve_log, " %.17g", blocks_[b][i]);
      std::fprintf(g_solve_log, " rel %.17g gmax %.17g\n", s.relative_decrease, s.gradient_max_norm);
    }
    return ceres::SOLVER_CONTINUE;
  }

 private:
  std::vector<double*> blocks_;
  std::vector<int> sizes_;
};
}  // namespace

namespace ceres {
// interposed: see the header comment.  Same signature as ceres/solver.h declares.
void Solve(const Solver::Options& options, Problem* problem, Solver::Summary* summary) {
  using Fn = void (*)(const Solver::Options&, Problem*, Solver::Summary*);
  static Fn real = reinterpret_cast<Fn>(dlsym(RTLD_NEXT, "_ZN5ceres5SolveERKNS_6Solver7OptionsEPNS_7ProblemEPNS0_7SummaryE"));
  if (!real) { std::fprintf(stderr, "ref_dump: the real ceres::Solve was not found behind this one (static libceres?)\n"); std::abort(); }
  std::vector<double*> blocks;
  problem->GetParameterBlocks(&blocks);                   // the reference has one: `parameters`, 6 doubles (registration.cpp:972-974)
  std::vector<int> sizes;
  for (double* b : blocks) sizes.push_back(problem->ParameterBlockSize(b));
  StateDump dump(blocks, sizes);
  Solver::Options with_dump = options;
  with_dump.update_state_every_iteration = true;
  with_dump.callbacks.push_back(&dump);
  real(with_dump, problem, summary);
  if (g_solve_log) {
    std::fprintf(g_solve_log, "s %d %d initial_cost %.17g final_cost %.17g successful %d unsuccessful %d termination %d\n", g_run,
                 g_call, summary->initial_cost, summary->final_cost, summary->num_successful_steps,
                 summary->num_unsuccessful_steps, static_cast<int>(summary->termination_type));
    // The quantities that DECIDE the recalled upstream behaviours the pose depends on (oracle/oracle_np.py ASSUMED_UPSTREAM;
    // tests/golden/assumption_sensitivity.json lists which ones move a pose by more than 1e-6 when flipped), by name -- "q" lines:
    //   residual_blocks / residuals    dist2_squared: registration.cpp:536 compares distance2[0] with 0.2 -- with squared distances
    //                                   (the assumption) more sphere blocks survive than with plain distances
    //   residual_evaluations / jacobian_evaluations   slot_after_solve: one cost sweep more than 1 + candidates means the minimiser
    //                                   evaluates once more after its loop, i.e. the side-channel slots hold the final state's costs;
    //                                   evaluate_on_add cannot be seen from inside Solve -- it shows as initial_cost of the NEXT
    //                                   outer iteration's Solve (the weights follow mu, which follows the slots at :1027-1033)
    //   iterations / successful / unsuccessful        iteration_count: max_num_iterations = 4 bounds all of them, or the successful ones
    //   termination / message          tolerance_exits: a Solve that ends by function / parameter tolerance with its last
    //                                   iteration's step_is_successful = 0 and the state unchanged ("x" lines) tested the
    //                                   tolerances before the step quality and did not apply the candidate
    //   (loss_correction shows in the iteration-1 numbers of the "i" lines: step_norm, cost_change, relative_decrease)
    std::fprintf(g_solve_log, "q %d %d residual_blocks %d residuals %d residual_evaluations %d jacobian_evaluations %d iterations %d "
                 "successful %d unsuccessful %d termination %d\n", g_run, g_call, problem->NumResidualBlocks(), problem->NumResiduals(),
                 summary->num_residual_evaluations, summary->num_jacobian_evaluations, static_cast<int>(summary->iterations.size()),
                 summary->num_successful_steps, summary->num_unsuccessful_steps, static_cast<int>(summary->termination_type));
    std::fprintf(g_solve_log, "m %d %d %s\n", g_run, g_call, summary->message.c_str());
    for (const IterationSummary& it : summary->iterations)
      std::fprintf(g_solve_log, "i %d %d %d cost %.17g change %.17g step_ok %d radius %.17g step_norm %.17g rel %.17g gmax %.17g valid %d\n",
                   g_run, g_call, it.iteration, it.cost, it.cost_change, it.step_is_successful ? 1 : 0, it.trust_region_radius,
                   it.step_norm, it.relative_decrease, it.gradient_max_norm, it.step_is_valid ? 1 : 0);
  }
  ++g_call;
}
}  // namespace ceres

namespace {
bool read_cloud(std::ifstream& f, std::shared_ptr<open3d::geometry::PointCloud2>& c) {
  int64_t n = 0;
  f.read(reinterpret_cast<char*>(&n), sizeof(n));
  if (!f || n < 0) return false;
  c->points_.resize(static_cast<size_t>(n));
  f.read(reinterpret_cast<char*>(c->points_.data()), static_cast<std::streamsize>(sizeof(double) * 3 * n));
  return static_cast<bool>(f);
}
YAML::Node tls_node(const double cfg[16], int max_iterations) {
  YAML::Node n;   // config/mapping/lidar_odometry.yaml:23-39, read by LocalRegistration::initConfig (registration.cpp:212-230)
  n["k_corr"] = static_cast<int>(cfg[0]);             n["factor_num"] = static_cast<int>(cfg[1]);
  n["edge_dist_thres"] = cfg[2];                      n["edge_dir_thres"] = cfg[3];
  n["edge_maxnum"] = static_cast<int>(cfg[4]);        n["sphere_maxnum"] = static_cast<int>(cfg[5]);
  n["sphere_dist_thres"] = cfg[6];                    n["planar_dist_thres"] = cfg[7];
  n["planar_maxnum"] = static_cast<int>(cfg[8]);      n["ground_maxnum"] = static_cast<int>(cfg[9]);
  n["ground_dist_thres"] = cfg[10];                   n["max_iterations"] = max_iterations;
  n["cost_threshold"] = cfg[12];                      n["gnc_factor"] = cfg[13];
  n["noise_bound"] = cfg[14];                         n["fitness_thres"] = cfg[15];
  return n;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: ref_dump <dir with case_*.bin> <output dir>\n");
    return 2;
  }
  const std::string in_dir = argv[1], out_dir = argv[2];
  int failures = 0;
  {
    // case list: the remaining arguments, or every case named in <in_dir>/cases.txt
    std::vector<std::string> cases;
    if (argc > 3) {
      for (int i = 3; i < argc; ++i) cases.push_back(argv[i]);
    } else {
      std::ifstream lst(in_dir + "/cases.txt");
      for (std::string s; std::getline(lst, s);)
        if (!s.empty()) cases.push_back(s);
    }
    for (const std::string& name : cases) {
      std::ifstream f(in_dir + "/" + name + ".bin", std::ios::binary);
      int32_t n_outer = 0;
      double cfg[16], Tp[16];
      f.read(reinterpret_cast<char*>(&n_outer), sizeof(n_outer));
      f.read(reinterpret_cast<char*>(cfg), sizeof(cfg));
      f.read(reinterpret_cast<char*>(Tp), sizeof(Tp));
      tloam::Frame src, tgt;
      // kind order of the C ABI: planar, ground, edge, sphere
      bool ok = static_cast<bool>(f) && read_cloud(f, src.planar_feature) && read_cloud(f, src.ground_feature) &&
                read_cloud(f, src.edge_feature) && read_cloud(f, src.sphere_feature) && read_cloud(f, tgt.planar_feature) &&
                read_cloud(f, tgt.ground_feature) && read_cloud(f, tgt.edge_feature) && read_cloud(f, tgt.sphere_feature);
      if (!ok) { std::fprintf(stderr, "%s: unreadable input\n", name.c_str()); ++failures; continue; }
      Eigen::Isometry3d predict = Eigen::Isometry3d::Identity();
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) predict.matrix()(r, c) = Tp[r * 4 + c];
      std::FILE* out = std::fopen((out_dir + "/" + name + ".ref.txt").c_str(), "w");
      if (!out) { ++failures; continue; }
      std::fprintf(out, "# %s: T_result (row-major 4x4) of the reference's scanMatching with max_iterations = k\n", name.c_str());
      g_solve_log = out;
      for (int k = 1; k <= n_outer; ++k) {
        g_run = k;
        g_call = 0;
        tloam::LocalRegistration reg(tls_node(cfg, k));          // registration.cpp:182-206
        reg.setInputSource(src);                                 // :232-239
        reg.setInputTarget(tgt);                                 // :241-248
        tloam::Frame result;                                     // empty scan cloud, as front_end.cpp:319 passes it
        Eigen::Isometry3d pose = Eigen::Isometry3d::Identity();
        reg.scanMatching(result, predict, pose);                 // :879-1133
        std::fprintf(out, "k %d", k);
        for (int r = 0; r < 4; ++r)
          for (int c = 0; c < 4; ++c) std::fprintf(out, " %.17g", pose.matrix()(r, c));
        std::fprintf(out, "\n");
      }
      g_solve_log = nullptr;
      std::fclose(out);
    }
  }
  return failures ? 1 : 0;
}
