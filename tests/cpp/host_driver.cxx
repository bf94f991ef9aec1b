// Test driver for the host C++ layer (wavenet::Wavenet / Coach / CostFunctions / operators).
// Reads doubles from a binary file, runs one scenario, prints named arrays as text
// ("name count v0 v1 ...", %.17g) for tests/test_host_cpp.py to compare with the oracle.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "Wavenet/Coach.h"
#include "Wavenet/Generators.h"
#include "Wavenet/Wavenet.h"
#include "wavenet_hip.h"

using namespace wavenet;

static std::vector<double> readAll(const char* path) {
    std::vector<double> v;
    FILE* f = std::fopen(path, "rb");
    if (!f) { std::perror(path); std::exit(2); }
    double buf[4096];
    size_t got;
    while ((got = std::fread(buf, sizeof(double), 4096, f)) > 0) v.insert(v.end(), buf, buf + got);
    std::fclose(f);
    return v;
}

static std::vector<int> deviceList(const char* text) {   // "0,1,3" -> {0, 1, 3}
    std::vector<int> out;
    for (const char* p = text; *p;) {
        char* end = nullptr;
        out.push_back((int)std::strtol(p, &end, 10));
        if (end == p) break;
        p = (*end == ',') ? end + 1 : end;
    }
    return out;
}

static std::string g_prefix;   // prepended to every dumped name
static void dump(const char* name, const double* p, size_t n) {
    std::printf("%s%s %zu", g_prefix.c_str(), name, n);
    for (size_t i = 0; i < n; i++) std::printf(" %.17g", p[i]);
    std::printf("\n");
}
static void dump(const char* name, const Mat& M) { dump(name, M.memptr(), M.n_elem); }
static void dumpLogs(Wavenet& w) {
    std::vector<double> flat;
    for (const Col& f : w.filterLog()) flat.insert(flat.end(), f.memptr(), f.memptr() + f.n_elem);
    dump("filterLog", flat.data(), flat.size());
    dump("costLog", w.costLog().data(), w.costLog().size());
    dump("filter", w.filter());
    dump("momentum", w.momentum());
    double misc[3] = {w.lambda(), w.alpha(), (double)w.batchSize()};
    dump("misc", misc, 3);
}

// exposes the protected transform methods, as a derived probe class would for the reference
struct Probe : public Wavenet {
    using Wavenet::backpropagate_;
    using Wavenet::forward_;
    using Wavenet::inverse_;
};

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const std::string mode = argv[1];
    if (mode == "ops") {
        const unsigned R = std::atoi(argv[2]), C = std::atoi(argv[3]), N = std::atoi(argv[4]);
        std::vector<double> in = readAll(argv[5]);
        const size_t n = (size_t)R * C;
        Probe w;
        if (!w.setFilter(Col(N, in.data()))) return 3;
        Mat X(R, C, in.data() + N), D(R, C, in.data() + N + n);
        Activations2D_t A = w.forward_(X);
        Mat Y = coeffsFromActivations(A);
        dump("Y", Y);
        dump("Xinv", w.inverse_(Y));
        dump("grad", w.backpropagate_(D, A));
        double c = w.cost(Y), s = SparseTerm(Y);
        dump("cost", &c, 1);
        dump("sparse", &s, 1);
        dump("sparseDeriv", SparseTermDeriv(Y));
        double r = RegTerm(w.filter());
        dump("reg", &r, 1);
        dump("regDeriv", RegTermDeriv(w.filter()));
        dump("lowpass", LowpassOperator(w.filter(), 2));
        dump("highpass", HighpassOperator(w.filter(), 2));
        if (C == 1) {
            Col x(R, X.memptr()), d(R, D.memptr());
            Activations1D_t a1 = w.forward_(x);
            dump("coeffs1d", coeffsFromActivations(a1));
            std::vector<Col> out = w.backpropagate_(d, a1);
            dump("grad1d", out[0]);
            dump("din1d", out[1]);
            dump("inv1d", w.inverse_(coeffsFromActivations(a1)));
        }
        dump("basis", w.basisFunction(R, C, R > 1 ? 1 : 0, C > 1 ? 1 : 0));
        {   // cost map over a 3 x 3 grid of 2-tap filters, two examples (X and D)
            const Col keep = w.filter();
            const size_t logged = w.filterLog().size();
            std::vector<Mat> cm = w.costMap({X, D}, 1.5, 3);
            dump("costmap", cm[0]);
            dump("costmapSparse", cm[1]);
            dump("costmapReg", cm[2]);
            double grew = (double)(w.filterLog().size() - logged);
            dump("costmapLogged", &grew, 1);
            w.setFilter(keep);
        }
        // error behaviour: odd filter and non radix-2 input are refused
        double flags[3] = {(double)w.setFilter(Col({1., 2., 3.})), (double)w.setFilter(Col()), (double)w.train(Mat(3, 1))};
        dump("refused", flags, 3);
        return 0;
    }
    if (mode == "sparse") {     // sparse n file: the free functions on a vector of any length
        const size_t n = (size_t)std::atoll(argv[2]);
        std::vector<double> in = readAll(argv[3]);
        const Col c(n, in.data());
        const double s = SparseTerm(c);
        dump("sparse", &s, 1);
        dump("sparseDeriv", SparseTermDeriv(c));
        return 0;
    }
    if (mode == "train") {
        const unsigned R = std::atoi(argv[2]), C = std::atoi(argv[3]), N = std::atoi(argv[4]), E = std::atoi(argv[5]);
        const unsigned batch = std::atoi(argv[6]);
        const double inertia = std::atof(argv[7]);   // "inertia" or "inertia:timescale"
        const char* colon = std::strchr(argv[7], ':');
        const double timeScale = colon ? std::atof(colon + 1) : 0.;
        const bool batched = std::atoi(argv[8]) != 0;
        std::vector<double> in = readAll(argv[9]);
        const size_t n = (size_t)R * C;
        Wavenet w(10.0, 1e-3);
        w.setBatchSize(batch);
        if (inertia > 0) w.setInertia(inertia);
        if (timeScale > 0) w.setInertiaTimeScale(timeScale);
        if (const char* rep = std::getenv("WAVENET_TEST_REPLICAS")) w.setDevices(std::vector<int>((size_t)std::atoi(rep), 0), false);
        if (const char* devs = std::getenv("WAVENET_TEST_DEVICES")) w.setDevices(deviceList(devs), std::getenv("WAVENET_TEST_RCCL") != nullptr);
        if (!w.setFilter(Col(N, in.data()))) return 3;
        if (batched) {
            // an odd split exercises the pending-queue path between device rounds
            const unsigned first = E / 3;
            if (!w.trainBatch(in.data() + N, first, R, C)) return 4;
            if (!w.trainBatch(in.data() + N + first * n, E - first, R, C)) return 4;
        } else {
            for (unsigned e = 0; e < E; e++) {
                if (!w.train(Mat(R, C, in.data() + N + e * n))) return 4;
            }
        }
        dumpLogs(w);
        return 0;
    }
    if (mode == "snapshot") {
        // load IN, dump; save to PATTERN with NUMBER (post-increment), load that file into a second
        // object, dump it with the prefix "re_".  No device call is made.
        const std::string in = argv[2], pattern = argv[3];
        const int number = std::atoi(argv[4]);
        Wavenet w;
        w.load(Snapshot(in));
        dumpLogs(w);
        Snapshot out(pattern, number);
        double exists[2] = {(double)out.exists(), 0};
        w.save(out++);
        Snapshot written(pattern, number);
        exists[1] = (double)written.exists();
        dump("exists", exists, 2);
        double num = out.number();
        dump("nextNumber", &num, 1);
        std::printf("file %s\n", written.file().c_str());
        Wavenet w2;
        w2.load(written);
        g_prefix = "re_";
        dumpLogs(w2);
        g_prefix = "";
        Wavenet w3(1.0, 2.0);                       // a missing file leaves the object alone
        w3.load(Snapshot(pattern + ".missing"));
        double kept[2] = {w3.lambda(), w3.alpha()};
        dump("missingKept", kept, 2);
        w3.save(Snapshot("/tmp/absolute_paths_are_refused.snap"));
        double refused = (double)fileExists("/tmp/absolute_paths_are_refused.snap");
        dump("absoluteWritten", &refused, 1);
        return 0;
    }
    if (mode == "resume") {
        // as "train" (one example per call), but the object is saved after `first` examples and a
        // fresh object, loaded from that file, takes over
        const unsigned R = std::atoi(argv[2]), C = std::atoi(argv[3]), N = std::atoi(argv[4]), E = std::atoi(argv[5]);
        const unsigned batch = std::atoi(argv[6]);
        const double inertia = std::atof(argv[7]);
        const unsigned first = std::atoi(argv[8]);
        const std::string file = argv[9];
        std::vector<double> in = readAll(argv[10]);
        const size_t n = (size_t)R * C;
        Wavenet w(10.0, 1e-3);
        w.setBatchSize(batch);
        if (inertia > 0) { w.setInertia(inertia); w.setInertiaTimeScale(5.); }
        if (!w.setFilter(Col(N, in.data()))) return 3;
        for (unsigned e = 0; e < first; e++) {
            if (!w.train(Mat(R, C, in.data() + N + e * n))) return 4;
        }
        w.save(Snapshot(file));
        Wavenet v;
        v.load(Snapshot(file));
        for (unsigned e = first; e < E; e++) {
            if (!v.train(Mat(R, C, in.data() + N + e * n))) return 4;
            if (!w.train(Mat(R, C, in.data() + N + e * n))) return 4;
        }
        dumpLogs(v);
        g_prefix = "straight_";
        dumpLogs(w);
        return 0;
    }
    if (mode == "coach") {
        const unsigned R = std::atoi(argv[2]), C = std::atoi(argv[3]), N = std::atoi(argv[4]);
        const int events = std::atoi(argv[5]);
        const unsigned epochs = std::atoi(argv[6]), batch = std::atoi(argv[7]), flags = std::atoi(argv[8]);
        const unsigned long long seed = std::strtoull(argv[9], nullptr, 10);
        const double target = std::atof(argv[10]);
        std::vector<double> in = readAll(argv[11]);
        Wavenet w(10.0, 1e-3);
        w.setBatchSize(batch);
        UniformGenerator gen;
        gen.setSeed(seed);
        if (!gen.setShape({R, C})) return 3;
        Coach coach("driver");
        coach.setWavenet(&w);
        coach.setGenerator(&gen);
        coach.setNumEvents(events);
        coach.setNumEpochs(epochs);
        coach.setNumCoeffs(N);
        coach.setInitialFilter(Col(N, in.data()));
        coach.setUseAdaptiveLearningRate(flags & 1u);
        coach.setUseAdaptiveBatchSize((flags >> 1) & 1u);
        coach.setUseSimulatedAnnealing((flags >> 2) & 1u);
        if (target > 0) coach.setTargetPrecision(target);
        if (const char* inits = std::getenv("WAVENET_TEST_NUM_INITS")) coach.setNumInits(std::atoi(inits));
        if (const char* rep = std::getenv("WAVENET_TEST_REPLICAS")) {
            // that many shares of every batch, all on GPU 0, their sums added on the host in replica order
            coach.setDevices(std::vector<int>((size_t)std::atoi(rep), 0), false);
        }
        if (const char* devs = std::getenv("WAVENET_TEST_DEVICES")) {
            // one share per listed GPU ("0,1"); sums through RCCL when WAVENET_TEST_RCCL is set, else added on the host
            coach.setDevices(deviceList(devs), std::getenv("WAVENET_TEST_RCCL") != nullptr);
        }
        if (const char* al = std::getenv("WAVENET_TEST_ALPHA")) w.setAlpha(std::atof(al));
        if (std::getenv("WAVENET_TEST_HOST_GENERATE")) coach.setGenerateOnDevice(false);
        if (const char* dir = std::getenv("WAVENET_TEST_SNAPSHOT_DIR")) {   // relative directory for the snapshots
            coach.setBasedir(dir);
            coach.setSaveSnapshots();
        }
        if (!coach.run()) return 4;
        double updates = (double)w.costLog().size();
        dump("updates", &updates, 1);
        dumpLogs(w);
        if (coach.saveSnapshots()) {   // what the snapshot of initialisation 0 holds
            Wavenet v;
            v.load(Snapshot(coach.outdir() + "snapshots/driver.%06u.snap", 0));
            g_prefix = "snap_";
            dumpLogs(v);
        }
        return 0;
    }
    if (mode == "generate") {
        // generate <kind> R C count seed: `count` examples from next() and the same from nextOnDevice, as "host" and "device"
        const std::string kind = argv[2];
        const unsigned R = std::atoi(argv[3]), C = std::atoi(argv[4]), count = std::atoi(argv[5]);
        const unsigned long long seed = std::strtoull(argv[6], nullptr, 10);
        UniformGenerator u; NeedleGenerator nd; GaussianGenerator ga; CalorimeterGenerator ca;
        SeededGenerator* gen = kind == "uniform" ? (SeededGenerator*)&u : kind == "needle" ? (SeededGenerator*)&nd
                             : kind == "gaussian" ? (SeededGenerator*)&ga : (SeededGenerator*)&ca;
        gen->setSeed(seed);
        if (!gen->setShape({R, C})) return 3;
        const size_t n = (size_t)R * C;
        std::vector<double> host(count * n), dev(count * n);
        for (unsigned e = 0; e < count; e++) std::memcpy(host.data() + e * n, gen->next().memptr(), n * sizeof(double));
        dump("host", host.data(), host.size());
        if (!std::getenv("WAVENET_TEST_HOST_ONLY")) {
            wn_context* ctx = nullptr;
            double* dX = nullptr;
            if (wn_create(0, &ctx) != WN_OK || wn_malloc(ctx, (void**)&dX, dev.size() * sizeof(double)) != WN_OK) return 4;
            gen->reset();
            if (!gen->nextOnDevice(ctx, dX, count)) return 5;
            if (wn_memcpy_d2h(ctx, dev.data(), dX, dev.size() * sizeof(double)) != WN_OK) return 6;
            dump("device", dev.data(), dev.size());
            wn_free(ctx, dX);
            wn_destroy(ctx);
        }
        return 0;
    }
    if (mode == "bench") {
        // bench R C N batch events epochs flags source kind: times Coach::run() (the reference's training entry point,
        // src/Coach.cxx:71-296) end to end and prints one JSON line.  source: "resident" (a fixed set of `events`
        // examples generated once on the device and served from there every epoch, as a file-based generator of the
        // reference would serve its file) or "generated" (every chunk drawn from the generator on the device).
        const unsigned R = std::atoi(argv[2]), C = std::atoi(argv[3]), N = std::atoi(argv[4]), batch = std::atoi(argv[5]);
        const int events = std::atoi(argv[6]);
        const unsigned epochs = std::atoi(argv[7]), flags = std::atoi(argv[8]);
        const std::string source = argv[9], kind = argv[10];
        std::vector<double> in = readAll(argv[11]);
        UniformGenerator u; NeedleGenerator nd; GaussianGenerator ga; CalorimeterGenerator ca;
        SeededGenerator* base = kind == "uniform" ? (SeededGenerator*)&u : kind == "needle" ? (SeededGenerator*)&nd
                              : kind == "gaussian" ? (SeededGenerator*)&ga : (SeededGenerator*)&ca;
        base->setSeed(1000);
        if (!base->setShape({R, C})) return 3;
        ResidentGenerator resident(*base, (unsigned)events);
        GeneratorBase* gen = source == "resident" ? (GeneratorBase*)&resident : (GeneratorBase*)base;
        double best = 0, updates = 0, lastCost = 0;
        std::vector<double> finalFilter;
        for (int rep = 0; rep < 3; rep++) {       // the first run also allocates, generates the resident set and loads the kernels
            Wavenet w(10.0, 1e-3);
            w.setBatchSize(batch);
            Coach coach("bench");
            coach.setWavenet(&w);
            coach.setGenerator(gen);
            coach.setNumEvents(source == "resident" ? -1 : events);
            coach.setNumEpochs(rep == 0 ? 1 : epochs);
            coach.setNumCoeffs(N);
            coach.setInitialFilter(Col(N, in.data()));
            coach.setUseAdaptiveLearningRate(flags & 1u);
            coach.setUseAdaptiveBatchSize((flags >> 1) & 1u);
            coach.setPrintLevel(0);
            struct timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            if (!coach.run()) return 4;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if (rep == 0) continue;
            const double s = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
            const double rate = (double)events * epochs / s;
            if (rate > best) best = rate;
            updates = (double)w.costLog().size();
            lastCost = w.lastCost();
            const Col f = w.filter();
            finalFilter.assign(f.memptr(), f.memptr() + N);
        }
        std::printf("{\"coach_run_examples_per_s\": %.1f, \"shape\": [%u, %u], \"taps\": %u, \"batch\": %u, \"events_per_epoch\": %d, "
                    "\"epochs\": %u, \"adaptive_flags\": %u, \"source\": \"%s\", \"generator\": \"%s\", \"updates\": %.0f, "
                    "\"last_cost\": %.12g, \"final_filter_0\": %.17g}\n",
                    best, R, C, N, batch, events, epochs, flags, source.c_str(), kind.c_str(), updates, lastCost,
                    finalFilter.empty() ? 0.0 : finalFilter[0]);
        return 0;
    }
    return 2;
}
