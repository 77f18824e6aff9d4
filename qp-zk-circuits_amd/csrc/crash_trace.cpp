// crash_trace.cpp — post-mortem evidence for a fatal signal inside a process that has libqpgpu loaded.
//
// Off unless the environment variable QPGPU_CRASH_TRACE is set when the library is loaded ("1" = report on stderr,
// anything else = path of a file to append to). On SIGSEGV / SIGBUS / SIGABRT / SIGFPE / SIGILL the handler writes,
// with async-signal-safe calls only: the signal, si_code and the faulting address, the kernel thread id, the line of
// /proc/self/maps that contains the address (tells a device BAR mapping, a pinned host buffer and ordinary heap apart),
// and the native backtrace of the faulting thread (glibc backtrace_symbols_fd: module + offset for every frame, enough
// for addr2line). It then hands over to whatever handler was installed before (python -X faulthandler dumps the Python
// stacks of all threads there) or re-raises with the default action, so exit status and core dumps are unchanged.
//
// Purpose: round 2 saw a host SIGSEGV in profiled (rocprofv3) multi-worker runs and kept no evidence of where. The
// profiling scripts under tools/gpurun_scripts/ set QPGPU_CRASH_TRACE so that a death that happens anyway leaves its
// backtrace under profiles/.
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace {

int g_fd = -1;
struct sigaction g_prev[NSIG];
const int kSignals[] = {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL};

void put(const char *s) { if (g_fd >= 0) { ssize_t r = write(g_fd, s, strlen(s)); (void)r; } }
void put_hex(uint64_t v) {
    char buf[19] = "0x0000000000000000";
    for (int i = 0; i < 16; i++) { const unsigned d = (unsigned)(v >> (60 - 4 * i)) & 15u; buf[2 + i] = (char)(d < 10 ? '0' + d : 'a' + d - 10); }
    put(buf);
}
void put_dec(long v) {
    char buf[24]; int n = 0;
    if (v < 0) { put("-"); v = -v; }
    do { buf[n++] = (char)('0' + v % 10); v /= 10; } while (v && n < 23);
    char out[25]; for (int i = 0; i < n; i++) out[i] = buf[n - 1 - i]; out[n] = 0;
    put(out);
}

uint64_t parse_hex(const char *&p) {
    uint64_t v = 0;
    for (;; p++) {
        const char ch = *p;
        unsigned d;
        if (ch >= '0' && ch <= '9') d = (unsigned)(ch - '0'); else if (ch >= 'a' && ch <= 'f') d = (unsigned)(ch - 'a' + 10); else break;
        v = (v << 4) | d;
    }
    return v;
}

// the /proc/self/maps line whose range holds `addr` (streamed through a small buffer: no allocation in a signal handler); when
// no mapping holds it, the nearest mapping below and the nearest above — that tells an access one past the end of a live buffer
// (the mapping below ends exactly at the address) from an access into a region that has been released (a gap on both sides)
void put_mapping_of(uint64_t addr) {
    const int fd = open("/proc/self/maps", O_RDONLY);
    if (fd < 0) { put("  (cannot read /proc/self/maps)\n"); return; }
    static char line[512], below[512], above[512];
    size_t len = 0;
    char ch;
    bool found = false, have_above = false;
    uint64_t below_hi = 0;
    below[0] = above[0] = 0;
    while (read(fd, &ch, 1) == 1) {
        if (ch != '\n') { if (len + 1 < sizeof line) line[len++] = ch; continue; }
        line[len] = 0;
        const char *p = line;
        const uint64_t lo = parse_hex(p);
        if (*p == '-') {
            p++;
            const uint64_t hi = parse_hex(p);
            if (addr >= lo && addr < hi) { put("  mapping: "); put(line); put("\n"); found = true; break; }
            if (hi <= addr) { memcpy(below, line, len + 1); below_hi = hi; }          // maps are sorted: the last one below wins
            else if (lo > addr && !have_above) { memcpy(above, line, len + 1); have_above = true; break; }
        }
        len = 0;
    }
    close(fd);
    if (found) return;
    put("  mapping: the address is not mapped in this process\n");
    if (below[0]) { put("    nearest mapping below: "); put(below); put(below_hi == addr ? "   <- ends exactly at the faulting address (one past its end)\n" : "\n");
                    if (below_hi != addr) { put("      gap between its end and the address: "); put_hex(addr - below_hi); put(" bytes\n"); } }
    if (above[0]) { put("    nearest mapping above: "); put(above); put("\n"); }
}

void handler(int sig, siginfo_t *si, void *uc) {
    static volatile sig_atomic_t busy = 0;
    if (!busy) {
        busy = 1;
        put("\n==== libqpgpu crash trace: signal "); put_dec(sig);
        put(sig == SIGSEGV ? " (SIGSEGV)" : sig == SIGBUS ? " (SIGBUS)" : sig == SIGABRT ? " (SIGABRT)" : sig == SIGFPE ? " (SIGFPE)" : " (SIGILL)");
        put(" si_code "); put_dec(si ? si->si_code : 0);
        put(" address "); put_hex(si ? (uint64_t)(uintptr_t)si->si_addr : 0);
        put(" thread "); put_dec((long)syscall(SYS_gettid));
        put(" pid "); put_dec((long)getpid()); put("\n");
        if (si && (sig == SIGSEGV || sig == SIGBUS)) put_mapping_of((uint64_t)(uintptr_t)si->si_addr);
        put("  native backtrace of the faulting thread:\n");
        void *frames[64];
        const int n = backtrace(frames, 64);
        if (g_fd >= 0) backtrace_symbols_fd(frames, n, g_fd);
        put("==== end of libqpgpu crash trace\n");
        busy = 0;
    }
    // hand over: the previous handler (e.g. faulthandler), or the default action
    const struct sigaction &prev = g_prev[sig];
    if ((prev.sa_flags & SA_SIGINFO) && prev.sa_sigaction) { prev.sa_sigaction(sig, si, uc); return; }
    if (!(prev.sa_flags & SA_SIGINFO) && prev.sa_handler != SIG_DFL && prev.sa_handler != SIG_IGN && prev.sa_handler) { prev.sa_handler(sig); return; }
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) void install() {
    const char *e = getenv("QPGPU_CRASH_TRACE");
    if (!e || !*e || (e[0] == '0' && !e[1])) return;
    g_fd = (e[0] == '1' && !e[1]) ? 2 : open(e, O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (g_fd < 0) g_fd = 2;
    { void *warm[4]; (void)backtrace(warm, 4); }   // loads libgcc's unwinder now: dlopen inside a signal handler is not safe
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    for (int sig : kSignals) sigaction(sig, &sa, &g_prev[sig]);
}

}  // namespace

// (exported so that a test can check the hook is linked in and see whether it is armed)
extern "C" int qpgpu_crash_trace_armed(void) { return g_fd >= 0 ? 1 : 0; }
