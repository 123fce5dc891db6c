// opal_amd [-n N] input.yaml -- the reference's command line, `mpirun -n N opal input.yaml` (src/main.rs:48-61), without MPI
// and without Python: the launcher forks N ranks BEFORE anything touches a GPU, one subdomain per rank and per GPU (ranks
// share a card when there are fewer cards than ranks), rank 0 writes the outputs next to the input file (main.rs:60-61).
//
// What MPI did for the reference is done here by
//   * the step's own kernels over peer-mapped mailboxes (opal_link_create_peer: the 64-byte IPC handles and the yes/no votes
//     of the set-up travel over Unix sockets made before the fork), else RCCL send/recv (the 128-byte unique id travels the
//     same way), else host-staged messages over the sockets between neighbours -- every rank ends up with the same kind;
//   * the output-time reductions (MPI_Reduce of the energies main.rs:23-31, the gather of the grid rows yee.rs:749-779, the
//     histogram all-reductions hgram.rs:190-192, 249, 260) over a star of sockets through rank 0.
// Environment: OPAL_RECUT=every[,threshold[,max_shift]] lets the cuts follow the particles (off by default), OPAL_LINK=peer|rccl|socket forces a transport, OPAL_LINK_CAPACITY the room for leavers per cut and step,
// OPAL_RCCL_LIB the RCCL to dlopen (default librccl.so).
#include <poll.h>
#include <signal.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/opal_amd.h"

namespace {

struct Comm {
    int rank = 0, size = 1;
    std::vector<int> hub;     // rank 0: hub[r] = socket to rank r (hub[0] unused); rank r > 0: hub[0] = socket to rank 0
    int nb[2] = {-1, -1};     // sockets to the left / right neighbour (ring)
};

bool write_all(int fd, const void *buf, size_t n)
{
    const char *p = static_cast<const char *>(buf);
    while (n) {
        ssize_t k = send(fd, p, n, MSG_NOSIGNAL);
        if (k < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += k;
        n -= (size_t)k;
    }
    return true;
}

bool read_all(int fd, void *buf, size_t n)
{
    char *p = static_cast<char *>(buf);
    while (n) {
        ssize_t k = recv(fd, p, n, 0);
        if (k <= 0) {
            if (k < 0 && errno == EINTR) continue;
            return false;  // the peer has gone
        }
        p += k;
        n -= (size_t)k;
    }
    return true;
}

// every rank contributes `n` bytes; every rank gets all of them in rank order
bool allgather(Comm &c, const void *mine, size_t n, std::vector<char> &all)
{
    all.assign(n * (size_t)c.size, 0);
    if (c.size == 1) {
        memcpy(all.data(), mine, n);
        return true;
    }
    if (c.rank == 0) {
        memcpy(all.data(), mine, n);
        for (int r = 1; r < c.size; r++)
            if (!read_all(c.hub[r], all.data() + n * (size_t)r, n)) return false;
        for (int r = 1; r < c.size; r++)
            if (!write_all(c.hub[r], all.data(), all.size())) return false;
        return true;
    }
    return write_all(c.hub[0], mine, n) && read_all(c.hub[0], all.data(), all.size());
}

bool all_agree(Comm &c, bool ok)
{
    char mine = ok ? 1 : 0;
    std::vector<char> all;
    if (!allgather(c, &mine, 1, all)) return false;
    return std::all_of(all.begin(), all.end(), [](char v) { return v != 0; });
}

// opal_allreduce_fn: in place over all subdomains, summed / compared in rank order on rank 0 (deterministic)
int cb_allreduce(void *user, double *v, size_t n, int op)
{
    Comm &c = *static_cast<Comm *>(user);
    std::vector<char> all;
    if (!allgather(c, v, n * sizeof(double), all)) return 1;
    const double *a = reinterpret_cast<const double *>(all.data());
    for (size_t i = 0; i < n; i++) {
        double acc = a[i];
        for (int r = 1; r < c.size; r++) {
            double x = a[(size_t)r * n + i];
            acc = op == 0 ? acc + x : op == 1 ? std::min(acc, x) : std::max(acc, x);
        }
        v[i] = acc;
    }
    return 0;
}

// opal_gather_fn: subdomain r contributes counts[r] doubles; rank 0 receives them concatenated in rank order
int cb_gather(void *user, const double *local, size_t n_local, double *all, const size_t *counts, int numtasks)
{
    Comm &c = *static_cast<Comm *>(user);
    if (c.rank != 0) return write_all(c.hub[0], local, n_local * sizeof(double)) ? 0 : 1;
    size_t at = 0;
    for (int r = 0; r < numtasks; r++) {
        if (r == 0) memcpy(all, local, n_local * sizeof(double));
        else if (!read_all(c.hub[r], all + at, counts[r] * sizeof(double))) return 1;
        at += counts[r];
    }
    return 0;
}

// opal_exchange2_fn over the neighbour sockets (host-staged messages): both sends and both receives progress together,
// so that neither end can block the other whatever the messages' sizes
int cb_exchange2(void *user, const void *send_left, const void *send_right, void *recv_left, void *recv_right,
                 size_t n_left, size_t n_right)
{
    Comm &c = *static_cast<Comm *>(user);
    struct Job { int fd; const char *src; char *dst; size_t ns, nr; } job[2] = {
        {c.nb[0], static_cast<const char *>(send_left), static_cast<char *>(recv_left), send_left ? n_left : 0, recv_left ? n_left : 0},
        {c.nb[1], static_cast<const char *>(send_right), static_cast<char *>(recv_right), send_right ? n_right : 0, recv_right ? n_right : 0}};
    while (job[0].ns || job[0].nr || job[1].ns || job[1].nr) {
        pollfd pf[2];
        for (int s = 0; s < 2; s++) {
            pf[s].fd = (job[s].ns || job[s].nr) ? job[s].fd : -1;
            pf[s].events = (short)((job[s].ns ? POLLOUT : 0) | (job[s].nr ? POLLIN : 0));
            pf[s].revents = 0;
        }
        if (poll(pf, 2, 120000) <= 0) return 1;  // a neighbour that says nothing for two minutes has gone
        for (int s = 0; s < 2; s++) {
            if (pf[s].revents & (POLLERR | POLLNVAL)) return 1;
            if (job[s].ns && (pf[s].revents & POLLOUT)) {
                ssize_t k = send(job[s].fd, job[s].src, job[s].ns, MSG_NOSIGNAL | MSG_DONTWAIT);
                if (k > 0) { job[s].src += k; job[s].ns -= (size_t)k; }
                else if (k < 0 && errno != EAGAIN && errno != EINTR) return 1;
            }
            if (job[s].nr && (pf[s].revents & (POLLIN | POLLHUP))) {
                ssize_t k = recv(job[s].fd, job[s].dst, job[s].nr, MSG_DONTWAIT);
                if (k > 0) { job[s].dst += k; job[s].nr -= (size_t)k; }
                else if (k == 0 || (errno != EAGAIN && errno != EINTR)) return 1;
            }
        }
    }
    return 0;
}
int cb_exchange(void *user, const void *sl, const void *sr, void *rl, void *rr, size_t n) { return cb_exchange2(user, sl, sr, rl, rr, n, n); }

struct LinkChoice {
    opal_link_t *link = nullptr;
    std::string kind;
};

// The neighbour transport, best first; collective: every rank ends up with the same kind (a kind is kept only if it came
// up on EVERY rank), or with an error.
bool make_link(Comm &c, void *stream, size_t cap, int slots, int fields, bool to_left, bool to_right, LinkChoice &out, std::string &why)
{
    const char *want = getenv("OPAL_LINK");
    auto allowed = [&](const char *k) { return !want || !*want || !strcmp(want, k); };
    if (allowed("peer")) {
        opal_link_t *l = nullptr;
        struct { char handle[OPAL_PEER_HANDLE_BYTES]; char ok; } mine;
        memset(&mine, 0, sizeof mine);
        mine.ok = opal_link_create_peer(&l, c.rank, c.size, cap, slots, fields, stream) == OPAL_OK &&
                  opal_link_peer_export(l, mine.handle) == OPAL_OK;
        if (!mine.ok) why = opal_last_error();
        std::vector<char> all;
        if (!allgather(c, &mine, sizeof mine, all)) return false;
        bool ok = true;
        for (int r = 0; r < c.size; r++) ok = ok && all[(size_t)r * sizeof mine + OPAL_PEER_HANDLE_BYTES];
        if (ok) {
            const char *left = all.data() + (size_t)((c.rank - 1 + c.size) % c.size) * sizeof mine;
            const char *right = all.data() + (size_t)((c.rank + 1) % c.size) * sizeof mine;
            bool mine_ok = opal_link_peer_connect(l, left, right) == OPAL_OK;
            if (!mine_ok) why = opal_last_error();
            ok = all_agree(c, mine_ok);  // every mailbox is mapped before anybody's first step writes into one
            if (ok) {
                mine_ok = opal_link_self_test(l, to_left, to_right, 8) == OPAL_OK;
                if (!mine_ok) why = opal_last_error();
                ok = all_agree(c, mine_ok);
            }
        }
        if (ok) {
            out.link = l;
            out.kind = "peer-mapped mailboxes";
            return true;
        }
        if (l) opal_link_free(l);
        if (c.rank == 0) fprintf(stderr, "peer link unavailable (%s); trying the next transport\n", why.c_str());
    }
    if (allowed("rccl")) {
        const char *lib = getenv("OPAL_RCCL_LIB");
        if (!lib || !*lib) lib = "librccl.so";
        struct { char id[OPAL_RCCL_UNIQUE_ID_BYTES]; char ok; } mine;
        memset(&mine, 0, sizeof mine);
        if (c.rank == 0) mine.ok = opal_rccl_unique_id(lib, mine.id) == OPAL_OK;
        std::vector<char> all;
        if (!allgather(c, &mine, sizeof mine, all)) return false;
        bool ok = all[OPAL_RCCL_UNIQUE_ID_BYTES] != 0;  // rank 0's
        opal_link_t *l = nullptr;
        if (ok) {
            bool mine_ok = opal_link_create_rccl(&l, c.rank, c.size, cap, slots, fields, lib, all.data(), stream) == OPAL_OK;
            if (!mine_ok) why = opal_last_error();
            ok = all_agree(c, mine_ok);
        } else {
            why = "no RCCL unique id on rank 0";
        }
        if (ok) {
            out.link = l;
            out.kind = "RCCL send/recv";
            return true;
        }
        if (l) opal_link_free(l);
        if (c.rank == 0) fprintf(stderr, "RCCL link unavailable (%s); trying the next transport\n", why.c_str());
    }
    if (allowed("socket")) {
        opal_link_t *l = nullptr;
        bool mine_ok = opal_link_create_layout(&l, c.rank, c.size, 0, cap, slots, fields, cb_exchange, &c, stream) == OPAL_OK &&
                       opal_link_set_exchange2(l, cb_exchange2) == OPAL_OK &&
                       opal_link_set_growth(l, getenv("OPAL_LINK_NO_GROWTH") ? 0 : 1) == OPAL_OK;  // (its buffers are the library's: link_grow)
        if (!mine_ok) why = opal_last_error();
        if (all_agree(c, mine_ok)) {
            out.link = l;
            out.kind = "host-staged messages over Unix sockets";
            return true;
        }
        if (l) opal_link_free(l);
    }
    if (why.empty()) why = "no neighbour transport could be set up";
    return false;
}

int run_rank(Comm &c, const char *path)
{
    auto fail = [&](const char *what) {
        fprintf(stderr, "Error%s: %s\n", c.size > 1 ? (" on rank " + std::to_string(c.rank)).c_str() : "", what);
        return 1;
    };
    int ndev = 0;
    if (opal_device_count(&ndev) != OPAL_OK) return fail(opal_last_error());
    if (opal_set_device(c.rank % ndev) != OPAL_OK) return fail(opal_last_error());
    void *stream = nullptr;
    if (opal_stream_create(&stream) != OPAL_OK) return fail(opal_last_error());
    LinkChoice lc;
    if (c.size > 1) {
        opal_config_t *cfg = nullptr;
        if (opal_config_from_file(&cfg, path) != OPAL_OK) return fail(opal_last_error());
        int emission = 0, absorption = 0;
        opal_config_read_bool(cfg, "qed", "photon_emission", &emission);
        opal_config_read_bool(cfg, "qed", "photon_absorption", &absorption);
        // room for the leavers of one step through one cut, per species, at creation: what stands within 8 cells of a cut on
        // both sides at the initial density (an edge grows from there when the particles near it require it: link_grow)
        size_t npc = 0, v = 0;
        for (const char *sec : {"electrons", "ions"})
            if (opal_config_contains(cfg, sec) == 1 && opal_config_read_usize(cfg, sec, "npc", &v) == OPAL_OK) npc = std::max(npc, v);
        size_t cap = std::max<size_t>(4096, 20 * npc);
        if (const char *e = getenv("OPAL_LINK_CAPACITY")) cap = (size_t)strtoull(e, nullptr, 10);
        const bool open_ends = opal_config_contains(cfg, "laser") == 1;  // laser + absorbing ends: one neighbour at each end
        opal_config_free(cfg);
        std::string why;
        if (!make_link(c, stream, cap, (emission || absorption) ? 3 : 2, absorption ? 15 : emission ? 12 : 6,
                       !(open_ends && c.rank == 0), !(open_ends && c.rank == c.size - 1), lc, why))
            return fail(why.c_str());
        if (c.rank == 0) {
            printf("Neighbour exchange: %s\n", lc.kind.c_str());
            fflush(stdout);
        }
    }
    opal_sim_t *sim = nullptr;
    int rc = opal_sim_create(&sim, path, c.rank, c.size, lc.link, stream);
    if (rc == OPAL_OK && c.size > 1) rc = opal_sim_set_collectives(sim, cb_allreduce, cb_gather, &c);
    if (rc == OPAL_OK && c.size > 1) {
        // OPAL_RECUT=every[,threshold[,max_shift]]: the cuts follow the particles (opal_sim_set_recut); unset: the reference's
        // partition, made once at t = 0 (grid/mod.rs:157-206)
        if (const char *e = getenv("OPAL_RECUT")) {
            unsigned long every = 0, shift = 64;
            double thr = 1.1;
            const int got = sscanf(e, "%lu,%lf,%lu", &every, &thr, &shift);
            if (got >= 1 && every) rc = opal_sim_set_recut(sim, every, thr, shift);
        }
    }
    if (rc == OPAL_OK && c.size > 1) {
        // the waits of the step's exchange are bounded: start together, however long each subdomain's loader took
        rc = opal_sim_sync(sim);
        if (!all_agree(c, rc == OPAL_OK) && rc == OPAL_OK) return fail("another subdomain failed to set up");
    }
    if (rc == OPAL_OK && c.rank == 0) {
        printf("Running %d task%s on %d GPU%s...\n", c.size, c.size > 1 ? "s" : "", std::min(c.size, ndev), std::min(c.size, ndev) > 1 ? "s" : "");
        fflush(stdout);
    }
    if (rc == OPAL_OK) rc = opal_sim_run(sim);
    if (rc != OPAL_OK) {
        // no collective from the error path: the other ranks are inside the run; the launcher ends them
        fail(opal_last_error());
        fflush(stderr);
        _exit(1);
    }
    if (c.size > 1 && !all_agree(c, true)) return fail("a subdomain did not finish");
    opal_sim_free(sim);
    if (lc.link) opal_link_free(lc.link);
    return 0;
}

}  // namespace

int main(int argc, char **argv)
{
    int n = 1;
    const char *path = nullptr;
    for (int i = 1; i < argc; i++) {
        if ((!strcmp(argv[i], "-n") || !strcmp(argv[i], "-np")) && i + 1 < argc) n = atoi(argv[++i]);
        else path = argv[i];
    }
    if (!path || n < 1) {
        fprintf(stderr, "Unable to open configuration file.\nUsage: opal_amd [-n ranks] input-file\n");  // main.rs:56-59
        return 1;
    }
    if (n == 1) {
        Comm c;
        return run_rank(c, path);
    }
    // sockets first, then the fork, and only the children touch a GPU (a forked copy of an initialised HIP runtime is
    // not usable, and the parent must stay able to end exactly the children it started)
    std::vector<int> hub0(n, -1), hubr(n, -1), ring_a(n, -1), ring_b(n, -1);
    for (int r = 1; r < n; r++) {
        int sv[2];
        if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv)) { perror("socketpair"); return 1; }
        hub0[r] = sv[0];
        hubr[r] = sv[1];
    }
    for (int r = 0; r < n; r++) {  // ring_a[r]: r's socket to its right neighbour; ring_b[r]: (r + 1) % n's socket to its left
        int sv[2];
        if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv)) { perror("socketpair"); return 1; }
        ring_a[r] = sv[0];
        ring_b[r] = sv[1];
    }
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);  // (the mailboxes' IPC handles need the dmabuf path)
    std::vector<pid_t> pids(n, -1);
    for (int r = 0; r < n; r++) {
        pid_t pid = fork();
        if (pid < 0) { perror("fork"); n = r; break; }
        if (pid == 0) {
            Comm c;
            c.rank = r;
            c.size = n;
            c.hub.assign(n, -1);
            for (int q = 1; q < n; q++) {
                if (r == 0) { c.hub[q] = hub0[q]; close(hubr[q]); }
                else if (q == r) { c.hub[0] = hubr[q]; close(hub0[q]); }
                else { close(hub0[q]); close(hubr[q]); }
            }
            for (int q = 0; q < n; q++) {
                bool keep_a = q == r, keep_b = (q + 1) % n == r;
                if (keep_a) c.nb[1] = ring_a[q]; else close(ring_a[q]);
                if (keep_b) c.nb[0] = ring_b[q]; else close(ring_b[q]);
            }
            int rc = run_rank(c, path);
            fflush(stdout);
            fflush(stderr);
            _exit(rc);
        }
        pids[r] = pid;
    }
    for (int r = 1; r < (int)hub0.size(); r++) { close(hub0[r]); close(hubr[r]); }
    for (size_t r = 0; r < ring_a.size(); r++) { close(ring_a[r]); close(ring_b[r]); }
    int failed = 0, left = n;
    while (left > 0) {
        int status = 0;
        pid_t pid = wait(&status);
        if (pid < 0) {
            if (errno == EINTR) continue;
            break;
        }
        left--;
        for (pid_t &p : pids)
            if (p == pid) p = -1;  // reaped: never signalled afterwards
        const bool bad = !(WIFEXITED(status) && WEXITSTATUS(status) == 0);
        if (bad && !failed) {
            failed = 1;
            for (pid_t p : pids)  // exactly the children started above
                if (p > 0) kill(p, SIGTERM);
        }
    }
    return failed;
}
