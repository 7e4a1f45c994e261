// oak_amd/csrc/pyoak_module.cc -- the Python face of the boundary: a pybind11 module with pyoak's names
// (cpp/src/pyoak.cc:428-716) over the C ABI of liboakgpu.so.  Import as `from oak_amd import pyoak`.
//
//   Heap(), Agent() {budget, bandit, eval, matrix_ucb, discrete, table}, Input()        pyoak.cc:442-454
//   parse_battle(battle_string, seed = 0x123456) -> Input                                 :456-466
//   update(input, c1, c2)                                                                 :468-478
//   Output {iterations, empirical_value, nash_value, duration_ms, visit_matrix[9,9], value_matrix[9,9],
//           p{1,2}_{prior,empirical,nash}[9]}                                             :494-574
//   search(input, heap, agent, output = Output()) -> Output                               :575-583
//   cpp_inference(record, network_path, discrete, budget) -> value / policy_logit / policy :331-392, 709
//   solve_matrix(row_payoff, discretize_factor) -> (p1, p2, value)                        :394-426, 711
//   read_battle_data(path) -> [(bytes, frame_count), ...]                                 :43-71, 713
//   network hyper-parameter constants                                                     :586-596
// Every battle operation goes through the C ABI (GPU); this file holds no battle arithmetic.  Not carried over: the
// training-data loaders of pyoak (EncodedBattleFrames, sample, BuildTrajectories): training stays with the reference's
// Python, outside the hot path.  Heap keeps the tree between searches (RuntimeSearch::Heap over oakgpu_heap; its C++
// `update(i, j, obs)` is exposed too, and `update(input, c1, c2)` returns the 16-byte observation that call needs), an
// `output` passed to search() is resumed like MCTS::Search::run's by-value Output (mcts.h:153-155), p{1,2}_prior are the
// softmax of the root's policy logits for contextual bandits (mcts.h:196-209).  cpp_inference takes a game record as
// read_battle_data returns it (the reference's BattleFrames loader is not carried over) and replays it exactly like
// pyoak.cc:331-392.  battle_string / parse_battle are served by the Python host mirror.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstring>
#include <fstream>
#include <mutex>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/oakgpu.h"

namespace py = pybind11;

namespace {

struct Input { // MCTS::Input (search/mcts.h:62-66): battle + public durations + result
  uint8_t battle[OAKGPU_BATTLE_SIZE] = {};
  uint8_t durations[OAKGPU_DURATIONS_SIZE] = {};
  uint8_t result = 0;
};
struct Heap { // RuntimeSearch::Heap (util/search.h:17-32): the tree of the last search, kept (oakgpu_heap)
  oakgpu_heap *h = nullptr;
  Heap() { if (oakgpu_heap_create(&h) != 0) throw std::runtime_error(oakgpu_last_error()); }
  Heap(const Heap &) = delete;
  Heap &operator=(const Heap &) = delete;
  ~Heap() { oakgpu_heap_destroy(h); }
  bool empty() const { return oakgpu_heap_empty(h) != 0; }
  std::string type() const {
    static const char *names[5] = {"UCB", "PUCB", "UCB1", "Exp3", "PExp3"};
    const int k = oakgpu_heap_kind(h);
    return k < 0 ? "std::monostate" : std::string("MCTS::Node<") + names[k] + "::JointBandit>";
  }
};
struct Agent { // RuntimeSearch::AgentParams (util/search.h:34-43)
  std::string budget = "4096", bandit = "ucb-1.0", eval = "mc", matrix_ucb;
  bool discrete = false, table = false;
};
struct Output { // MCTS::Output (search/mcts.h:68-90) as pyoak exposes it
  oakgpu_search_output raw{};
};

void check(int rc) {
  if (rc != 0) throw std::runtime_error(oakgpu_last_error());
}

std::mutex g_ctx_mu; // a context serves one caller at a time; search() drops the GIL, so two Python threads could meet here
oakgpu_ctx *context() { // one context per process, created on first use (device from OAKGPU_DEVICE, default 0)
  static oakgpu_ctx *ctx = nullptr;
  if (!ctx) {
    const char *env = std::getenv("OAKGPU_DEVICE");
    check(oakgpu_create(&ctx, env ? std::atoi(env) : 0));
  }
  return ctx;
}

template <class T, class F> py::array_t<T> vec9(F f) {
  py::array_t<T> arr(9);
  auto r = arr.template mutable_unchecked<1>();
  for (py::ssize_t i = 0; i < 9; ++i) r(i) = f((int)i);
  return arr;
}

} // namespace

PYBIND11_MODULE(pyoak, m) {
  m.doc() = "pyoak-compatible bindings over liboakgpu.so (MI355X)";

  py::class_<Heap>(m, "Heap")
      .def(py::init<>())
      .def("empty", &Heap::empty)
      .def("type", &Heap::type)
      // RuntimeSearch::Heap::update(i, j, obs) (search.cc:27-52; not bound by pyoak, used by its C++ callers vs.cc / chall.cc)
      .def("update", [](Heap &hp, int i, int j, py::bytes obs) {
             const std::string o = obs;
             if (o.size() != 16) throw std::runtime_error("Heap.update: obs must be 16 bytes");
             return oakgpu_heap_update(hp.h, (uint8_t)i, (uint8_t)j, (const uint8_t *)o.data()) != 0;
           }, py::arg("i"), py::arg("j"), py::arg("obs"))
      .def("nodes", [](const Heap &hp) { return (size_t)oakgpu_heap_nodes(hp.h); });

  py::class_<Agent>(m, "Agent")
      .def(py::init<>())
      .def_readwrite("budget", &Agent::budget)
      .def_readwrite("bandit", &Agent::bandit)
      .def_readwrite("eval", &Agent::eval)
      .def_readwrite("matrix_ucb", &Agent::matrix_ucb)
      .def_readwrite("discrete", &Agent::discrete)
      .def_readwrite("table", &Agent::table);

  py::class_<Input>(m, "Input")
      .def(py::init<>())
      // not in pyoak (its Input is opaque): raw views for tests and for feeding the batched C ABI
      .def_property_readonly("battle", [](const Input &i) { return py::bytes((const char *)i.battle, sizeof i.battle); })
      .def_property_readonly("durations", [](const Input &i) { return py::bytes((const char *)i.durations, sizeof i.durations); })
      .def_property_readonly("result", [](const Input &i) { return (int)i.result; });

  m.def(
      "parse_battle",
      [](const std::string &battle_string, uint64_t seed) {
        // Parse::parse_battle (util/parse.h:14-282) lives in the Python host mirror (oak_amd/parse.py)
        py::object mod = py::module_::import("oak_amd.parse");
        py::tuple bd = mod.attr("parse_battle")(battle_string, seed).cast<py::tuple>();
        auto b = bd[0].cast<py::array_t<uint8_t, py::array::c_style | py::array::forcecast>>();
        auto d = bd[1].cast<py::array_t<uint8_t, py::array::c_style | py::array::forcecast>>();
        if (b.size() != OAKGPU_BATTLE_SIZE || d.size() != OAKGPU_DURATIONS_SIZE) throw std::runtime_error("parse_battle: bad array sizes");
        Input in;
        std::memcpy(in.battle, b.data(), sizeof in.battle);
        std::memcpy(in.durations, d.data(), sizeof in.durations);
        in.result = mod.attr("result_from_state")(bd[0]).cast<uint8_t>(); // PKMN::result(battle), pkmn.h:235-272
        return in;
      },
      py::arg("battle_string"), py::arg("seed") = 0x123456);

  m.def(
      "update",
      [](Input &input, uint8_t c1, uint8_t c2) { // options <- durations; PKMN::update; durations <- options (pyoak.cc:468-478)
        uint8_t actions[16];
        std::lock_guard<std::mutex> lock(g_ctx_mu);
        check(oakgpu_update(context(), input.battle, &c1, &c2, input.durations, actions, nullptr, 1, &input.result));
        return py::bytes((const char *)actions, 16); // the observation of this update (pyoak returns None): Heap.update's obs
      },
      py::arg("input"), py::arg("c1"), py::arg("c2"));

  m.def(
      "choices",
      [](const Input &input) { // not in pyoak: PKMN::choices(battle, result) (pkmn.h:141-156) for both players
        uint8_t c1[9], c2[9], n1 = 0, n2 = 0;
        std::lock_guard<std::mutex> lock(g_ctx_mu);
        check(oakgpu_choices(context(), input.battle, &input.result, 0, c1, &n1, 1));
        check(oakgpu_choices(context(), input.battle, &input.result, 1, c2, &n2, 1));
        return py::make_tuple(std::vector<int>(c1, c1 + n1), std::vector<int>(c2, c2 + n2));
      },
      py::arg("input"));

  m.def(
      "battle_string",
      [](const Input &input) {
        py::object mod = py::module_::import("oak_amd.parse");
        return mod.attr("battle_string")(py::bytes((const char *)input.battle, 384), py::bytes((const char *)input.durations, 8)).cast<std::string>();
      },
      py::arg("input"));

  py::class_<Output>(m, "Output")
      .def(py::init<>())
      .def_property_readonly("iterations", [](const Output &o) { return o.raw.iterations; })
      .def_property_readonly("empirical_value", [](const Output &o) { return o.raw.empirical_value; })
      .def_property_readonly("nash_value", [](const Output &o) { return o.raw.nash_value; })
      .def_property_readonly("duration_ms", [](const Output &o) { return o.raw.duration_us / 1e3; })
      .def_property_readonly("m", [](const Output &o) { return (int)o.raw.m; })
      .def_property_readonly("n", [](const Output &o) { return (int)o.raw.n; })
      .def_property_readonly("p1_choices", [](const Output &o) { return std::vector<int>(o.raw.p1_choices, o.raw.p1_choices + o.raw.m); })
      .def_property_readonly("p2_choices", [](const Output &o) { return std::vector<int>(o.raw.p2_choices, o.raw.p2_choices + o.raw.n); })
      .def_property_readonly("visit_matrix",
                             [](const Output &o) {
                               py::array_t<size_t> arr({9, 9});
                               auto r = arr.mutable_unchecked<2>();
                               for (int i = 0; i < 9; ++i)
                                 for (int j = 0; j < 9; ++j) r(i, j) = (i < o.raw.m && j < o.raw.n) ? o.raw.visit_matrix[i * 9 + j] : 0;
                               return arr;
                             })
      .def_property_readonly("value_matrix",
                             [](const Output &o) {
                               py::array_t<double> arr({9, 9});
                               auto r = arr.mutable_unchecked<2>();
                               for (int i = 0; i < 9; ++i)
                                 for (int j = 0; j < 9; ++j) r(i, j) = (i < o.raw.m && j < o.raw.n) ? o.raw.value_matrix[i * 9 + j] : 0.0;
                               return arr;
                             })
      .def_property_readonly("initial_value", [](const Output &o) { return o.raw.initial_value; })
      .def_property_readonly("p1_logit", [](const Output &o) { return vec9<double>([&](int i) { return o.raw.p1_logit[i]; }); })
      .def_property_readonly("p2_logit", [](const Output &o) { return vec9<double>([&](int i) { return o.raw.p2_logit[i]; }); })
      .def_property_readonly("p1_prior", [](const Output &o) { return vec9<double>([&](int i) { return o.raw.p1_prior[i]; }); })
      .def_property_readonly("p2_prior", [](const Output &o) { return vec9<double>([&](int i) { return o.raw.p2_prior[i]; }); })
      .def_property_readonly("p1_empirical", [](const Output &o) { return vec9<double>([&](int i) { return o.raw.p1_empirical[i]; }); })
      .def_property_readonly("p2_empirical", [](const Output &o) { return vec9<double>([&](int i) { return o.raw.p2_empirical[i]; }); })
      .def_property_readonly("p1_nash", [](const Output &o) { return vec9<double>([&](int i) { return o.raw.p1_nash[i]; }); })
      .def_property_readonly("p2_nash", [](const Output &o) { return vec9<double>([&](int i) { return o.raw.p2_nash[i]; }); });

  m.def(
      "format",
      [](const Input &input, const Output &o) { // MCTS::output_string(output, input), pyoak.cc:487-492
        py::object mod = py::module_::import("oak_amd.parse");
        py::dict d;
        const int mm = o.raw.m, nn = o.raw.n;
        d["m"] = mm; d["n"] = nn;
        d["iterations"] = (uint64_t)o.raw.iterations;
        d["duration_ms"] = o.raw.duration_us / 1e3;
        d["empirical_value"] = o.raw.empirical_value;
        d["p1_choices"] = std::vector<int>(o.raw.p1_choices, o.raw.p1_choices + mm);
        d["p2_choices"] = std::vector<int>(o.raw.p2_choices, o.raw.p2_choices + nn);
        d["p1_empirical"] = std::vector<double>(o.raw.p1_empirical, o.raw.p1_empirical + 9);
        d["p2_empirical"] = std::vector<double>(o.raw.p2_empirical, o.raw.p2_empirical + 9);
        d["p1_nash"] = std::vector<double>(o.raw.p1_nash, o.raw.p1_nash + 9);
        d["p2_nash"] = std::vector<double>(o.raw.p2_nash, o.raw.p2_nash + 9);
        d["p1_prior"] = std::vector<double>(o.raw.p1_prior, o.raw.p1_prior + 9);
        d["p2_prior"] = std::vector<double>(o.raw.p2_prior, o.raw.p2_prior + 9);
        py::array_t<double> vis({9, 9}), val({9, 9});
        auto rv = vis.mutable_unchecked<2>(); auto rw = val.mutable_unchecked<2>();
        for (int i = 0; i < 9; ++i)
          for (int j = 0; j < 9; ++j) { rv(i, j) = (double)o.raw.visit_matrix[i * 9 + j]; rw(i, j) = o.raw.value_matrix[i * 9 + j]; }
        d["visit_matrix"] = vis; d["value_matrix"] = val;
        return mod.attr("format_output")(py::bytes((const char *)input.battle, 384), d).cast<std::string>();
      },
      py::arg("input"), py::arg("output"));

  m.def(
      "search",
      [](const Input &input, Heap &heap, Agent &agent, Output previous, uint32_t batch, py::object seed) {
        const oakgpu_agent a{agent.budget.c_str(), agent.bandit.c_str(), agent.eval.c_str(), agent.matrix_ucb.c_str(), agent.discrete, agent.table};
        // pyoak seeds its device from std::random_device on every call (pyoak.cc:579); a seed may be given for reproducibility
        const uint64_t s = seed.is_none() ? ((uint64_t)std::random_device{}() << 32) ^ std::random_device{}() : seed.cast<uint64_t>();
        Output out;
        int rc;
        {
          py::gil_scoped_release release; // the search is long and touches no Python state
          std::lock_guard<std::mutex> lock(g_ctx_mu);
          // RuntimeSearch::run(device, input, heap, agent, output) (pyoak.cc:575-583): the heap's tree and `output` are resumed
          rc = oakgpu_search_agent_heap(context(), heap.h, input.battle, input.durations, input.result, &a, batch, s, &previous.raw, &out.raw);
        }
        check(rc);
        return out;
      },
      py::arg("input"), py::arg("heap"), py::arg("agent"), py::arg("output") = Output{}, py::arg("batch") = 0, py::arg("seed") = py::none());

  m.def(
      "value_policy_inference",
      [](const Input &input, const std::string &network_path) {
        // NetworkImpl::value_policy_inference at one position (network.h:102-123), the way Search::run calls it for a fresh
        // root of a contextual bandit (mcts.h:196-209): a zero-iteration "pucb" search leaves value + legal logits in the output
        const oakgpu_agent a{"0", "pucb-1.0", network_path.c_str(), "", 0, 0};
        oakgpu_search_output out{};
        {
          std::lock_guard<std::mutex> lock(g_ctx_mu);
          check(oakgpu_search_agent_heap(context(), nullptr, input.battle, input.durations, input.result, &a, 1, 0, nullptr, &out));
        }
        py::array_t<float> l1(out.m), l2(out.n);
        for (int i = 0; i < out.m; ++i) l1.mutable_unchecked<1>()(i) = (float)out.p1_logit[i];
        for (int j = 0; j < out.n; ++j) l2.mutable_unchecked<1>()(j) = (float)out.p2_logit[j];
        return py::make_tuple((float)out.initial_value, l1, l2);
      },
      py::arg("input"), py::arg("network_path"));
  m.def(
      "value_inference",
      [](const Input &input, const std::string &network_path) { // NetworkImpl::value_inference (network.h:72-79)
        const oakgpu_agent a{"0", "pucb-1.0", network_path.c_str(), "", 0, 0};
        oakgpu_search_output out{};
        std::lock_guard<std::mutex> lock(g_ctx_mu);
        check(oakgpu_search_agent_heap(context(), nullptr, input.battle, input.durations, input.result, &a, 1, 0, nullptr, &out));
        return (float)out.initial_value;
      },
      py::arg("input"), py::arg("network_path"));

  m.def(
      "cpp_inference",
      [](py::bytes record, const std::string &network_path, bool discrete, const std::string &budget) {
        // pyoak.cc:331-392 (the C++ side of the reference's torch == C++ check, src/oak/lab.py:21-79): replay one game record;
        // at every frame run RuntimeSearch::run with agent {eval = network, bandit = "pucb-1.0", budget} on a fresh heap and
        // keep initial_value, the legal logits and their softmax; then play the stored choices.  Like the reference, the
        // first frame starts from the record's battle with zero durations and result None|Move|Move.
        const std::string rec = record;
        uint8_t battle[384], final_result = 0;
        uint32_t count = 0;
        check(oakgpu_frames_read((const uint8_t *)rec.data(), rec.size(), battle, &final_result, nullptr, 0, &count, nullptr));
        std::vector<oakgpu_frame_update> ups(count ? count : 1);
        check(oakgpu_frames_read((const uint8_t *)rec.data(), rec.size(), nullptr, nullptr, ups.data(), count, &count, nullptr));
        py::array_t<float> value(count), logit({(py::ssize_t)count, (py::ssize_t)2, (py::ssize_t)9}), policy({(py::ssize_t)count, (py::ssize_t)2, (py::ssize_t)9});
        auto v = value.mutable_unchecked<1>();
        auto lg = logit.mutable_unchecked<3>();
        auto po = policy.mutable_unchecked<3>();
        uint8_t durations[8] = {}, result = 0x50; // PKMN::result(): None, p1 Move, p2 Move (pkmn.h:228-233)
        const oakgpu_agent a{budget.c_str(), "pucb-1.0", network_path.c_str(), "", discrete ? 1 : 0, 0};
        std::lock_guard<std::mutex> lock(g_ctx_mu);
        for (uint32_t f = 0; f < count; ++f) {
          oakgpu_search_output out{};
          check(oakgpu_search_agent_heap(context(), nullptr, battle, durations, result, &a, 0, std::random_device{}(), nullptr, &out));
          v(f) = (float)out.initial_value;
          for (int q = 0; q < 9; ++q) {
            lg(f, 0, q) = q < out.m ? (float)out.p1_logit[q] : 0.0f;
            lg(f, 1, q) = q < out.n ? (float)out.p2_logit[q] : 0.0f;
            po(f, 0, q) = q < out.m ? (float)out.p1_prior[q] : 0.0f;
            po(f, 1, q) = q < out.n ? (float)out.p2_prior[q] : 0.0f;
          }
          check(oakgpu_update(context(), battle, &ups[f].c1, &ups[f].c2, durations, nullptr, nullptr, 1, &result));
        }
        py::dict d;
        d["value"] = value;
        d["policy_logit"] = logit;
        d["policy"] = policy;
        return d;
      },
      py::arg("record"), py::arg("network_path"), py::arg("discrete") = false, py::arg("budget") = "0");

  m.def(
      "solve_matrix",
      [](py::array_t<float> p1_payoffs, int discretize_factor) { // pyoak.cc:394-426: float payoffs, discretised here
        if (p1_payoffs.ndim() != 2) throw std::runtime_error{"Expecting 2d array"};
        const auto mm = p1_payoffs.shape(0), nn = p1_payoffs.shape(1);
        auto r = p1_payoffs.unchecked<2>();
        std::vector<int32_t> disc((size_t)(mm * nn));
        for (py::ssize_t i = 0; i < mm; ++i)
          for (py::ssize_t j = 0; j < nn; ++j) disc[(size_t)(i * nn + j)] = static_cast<int>(r(i, j) * discretize_factor);
        std::vector<double> a((size_t)std::max<py::ssize_t>(mm, 1)), b((size_t)std::max<py::ssize_t>(nn, 1));
        double value = 0;
        check(oakgpu_solve_matrix(disc.data(), (int)mm, (int)nn, discretize_factor, a.data(), b.data(), &value));
        py::array_t<float> p1(mm), p2(nn);
        for (py::ssize_t i = 0; i < mm; ++i) p1.mutable_unchecked<1>()(i) = (float)a[(size_t)i];
        for (py::ssize_t j = 0; j < nn; ++j) p2.mutable_unchecked<1>()(j) = (float)b[(size_t)j];
        return py::make_tuple(p1, p2, (float)value);
      },
      py::arg("row_payoff"), py::arg("discretize_factor") = 256);

  m.def(
      "read_battle_data",
      [](const std::string &path) { // pyoak.cc:43-71: [(record bytes, frame count), ...]
        std::ifstream file(path, std::ios::binary);
        if (!file) throw std::runtime_error("read_battle_data: Failed to open file: " + path);
        std::vector<char> data((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
        py::list result;
        size_t pos = 0;
        while (pos < data.size()) {
          uint32_t count = 0;
          size_t used = 0;
          if (oakgpu_frames_read((const uint8_t *)data.data() + pos, data.size() - pos, nullptr, nullptr, nullptr, 0, &count, &used) != 0)
            throw std::runtime_error(std::string("read_battle_data: ") + oakgpu_last_error());
          result.append(py::make_tuple(py::bytes(data.data() + pos, used), (int)count));
          pos += used;
        }
        return result;
      },
      py::arg("path"));

  // Battle net hyper-parameters (pyoak.cc:586-596; nn/default-hyperparameters.h:10-18, encode/battle/*.h dims)
  m.attr("pokemon_in_dim") = 198;
  m.attr("active_in_dim") = 427;
  m.attr("pokemon_hidden_dim") = 128;
  m.attr("pokemon_out_dim") = 59;
  m.attr("active_hidden_dim") = 128;
  m.attr("active_out_dim") = 83;
  m.attr("side_out_dim") = 384;
  m.attr("hidden_dim") = 64;
  m.attr("value_hidden_dim") = 32;
  m.attr("policy_hidden_dim") = 64;
  m.attr("policy_out_dim") = 315;
}
