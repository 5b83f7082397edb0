/* Test infrastructure: native backtrace of the thread that dies.
 *
 * Loaded into the pytest process (tests/conftest.py) and into the stress scripts: when the process
 * receives SIGABRT / SIGSEGV / SIGBUS / SIGFPE / SIGILL the handler writes the C-level stack of the
 * RAISING thread (abort() runs the handler on the thread that called it; a fault is delivered to the
 * faulting thread) to fd 2 with backtrace_symbols_fd() - no malloc, usable after heap corruption -
 * plus /proc/self/maps lines of the HIP / HSA / library objects so that offsets can be resolved, then
 * chains to the previous handler (Python's faulthandler) or re-raises with the default action.
 * Python's faulthandler only shows the Python frames; "Aborted (core dumped)" inside a ctypes call
 * says nothing about WHO called abort(). Not part of the product.
 */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static struct sigaction g_prev[65];

static void put(const char* s) { (void)!write(2, s, strlen(s)); }

static void dump_maps(void) {
    /* lines of /proc/self/maps that name an executable mapping of a library we care about */
    static char buf[1 << 16];
    int fd = open("/proc/self/maps", O_RDONLY);
    if (fd < 0) return;
    size_t have = 0;
    for (;;) {
        ssize_t n = read(fd, buf + have, sizeof(buf) - 1 - have);
        if (n <= 0) break;
        have += (size_t)n;
        buf[have] = 0;
        char* line = buf;
        for (;;) {
            char* nl = strchr(line, '\n');
            if (!nl) break;
            *nl = 0;
            if (strstr(line, " r-xp ") &&
                (strstr(line, "libamdhip64") || strstr(line, "libhsa-runtime") || strstr(line, "libcgvec_hip") ||
                 strstr(line, "librccl") || strstr(line, "libc.so") || strstr(line, "libstdc++") ||
                 strstr(line, "libtorch_hip") || strstr(line, "libc10_hip"))) {
                put("  map ");
                put(line);
                put("\n");
            }
            line = nl + 1;
        }
        have = strlen(line);
        memmove(buf, line, have);
    }
    close(fd);
}

static void handler(int sig, siginfo_t* info, void* uc) {
    static volatile int entered = 0;
    if (!__sync_lock_test_and_set(&entered, 1)) {
        void* frames[96];
        put("\n==== abort_bt: native backtrace of the raising thread (signal ");
        char num[4] = {(char)('0' + sig / 10), (char)('0' + sig % 10), 0, 0};
        put(num);
        put(") ====\n");
        int n = backtrace(frames, 96);
        backtrace_symbols_fd(frames, n, 2);
        dump_maps();
        put("==== abort_bt: end ====\n");
    }
    struct sigaction* p = &g_prev[sig];
    if ((p->sa_flags & SA_SIGINFO) && p->sa_sigaction) {
        p->sa_sigaction(sig, info, uc);
        return;
    }
    if (!(p->sa_flags & SA_SIGINFO) && p->sa_handler != SIG_DFL && p->sa_handler != SIG_IGN) {
        p->sa_handler(sig);
        return;
    }
    signal(sig, SIG_DFL);
    raise(sig);
}

/* install (idempotent); returns the number of signals hooked */
int abort_bt_install(void) {
    static int done = 0;
    if (done) return 0;
    done = 1;
    void* warm[4];
    (void)backtrace(warm, 4); /* loads libgcc now, not inside the handler */
    const int sigs[] = {SIGABRT, SIGSEGV, SIGBUS, SIGFPE, SIGILL};
    int k = 0;
    for (unsigned i = 0; i < sizeof(sigs) / sizeof(sigs[0]); ++i) {
        struct sigaction sa;
        memset(&sa, 0, sizeof(sa));
        sa.sa_sigaction = handler;
        sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
        sigemptyset(&sa.sa_mask);
        if (sigaction(sigs[i], &sa, &g_prev[sigs[i]]) == 0) ++k;
    }
    return k;
}
