/* TEST INFRASTRUCTURE (tests/conftest.py, GPU sessions only): a native backtrace when the test process dies of SIGABRT.
 * Round 5 saw one abort inside torch's tensor.to(device) that left nothing but Python frames (faulthandler) -- glibc's heap
 * checks, ROCclr's guarantee() and the HSA runtime all end in abort(), and the frames above it say which it was.  The handler
 * writes the frames (backtrace_symbols_fd: async-signal-safe, no malloc) to the descriptors it was given, puts the default
 * action back and returns, so that abort() goes on to end the process as it would have. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static int g_fd[2] = {-1, -1};

static void on_abort(int sig)
{
    void *frames[96];
    const int n = backtrace(frames, 96);
    static const char head[] = "\n--- native frames at SIGABRT (tests/abort_trace.c) ---\n";
    for (int k = 0; k < 2; ++k) {
        if (g_fd[k] < 0) continue;
        if (write(g_fd[k], head, sizeof head - 1) < 0) continue;
        backtrace_symbols_fd(frames, n, g_fd[k]);
    }
    /* What native code printed to stderr while the dying test ran -- glibc's "free(): invalid pointer", the HSA runtime's "Memory
     * access fault by GPU node", ROCclr's queue error callback -- went into pytest's capture file (descriptor 2 is that temporary
     * file during a test, opened read-write) and would die with the process: its tail is copied to the same places. */
    if (g_fd[0] >= 0 || g_fd[1] >= 0) {
        static char buf[8192];
        static const char head2[] = "--- stderr captured during the dying test (tail) ---\n";
        const off_t end = lseek(2, 0, SEEK_CUR);
        if (end > 0) {
            off_t at = end > (off_t)(8 * sizeof buf) ? end - (off_t)(8 * sizeof buf) : 0;
            for (int k = 0; k < 2; ++k)
                if (g_fd[k] >= 0 && write(g_fd[k], head2, sizeof head2 - 1) < 0) g_fd[k] = -1;
            while (at < end) {
                const ssize_t n = pread(2, buf, sizeof buf < (size_t)(end - at) ? sizeof buf : (size_t)(end - at), at);
                if (n <= 0) break;
                for (int k = 0; k < 2; ++k)
                    if (g_fd[k] >= 0 && write(g_fd[k], buf, (size_t)n) < 0) g_fd[k] = -1;
                at += n;
            }
        }
    }
    signal(sig, SIG_DFL);
}

/* fd_a / fd_b: where to write (-1: nowhere).  Returns 0. */
int xm_install_abort_trace(int fd_a, int fd_b)
{
    void *warm[4];
    (void)backtrace(warm, 4);            /* loads libgcc's unwinder now, not inside the handler */
    g_fd[0] = fd_a;
    g_fd[1] = fd_b;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_abort;
    sigemptyset(&sa.sa_mask);
    return sigaction(SIGABRT, &sa, NULL);
}
