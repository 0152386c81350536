// In-library launch timing for bench.py's roofline legs (HIP events on the launch stream, per kernel class).  Off unless tcl_prof_begin()
// switched a class on; a scope costs one branch then.  Events come from a free list and are resolved lazily (a 300-frame pass has ~1e5
// launches per class: the live event count stays bounded).
#pragma once
#include <hip/hip_runtime.h>
#include <deque>
#include <vector>

enum { TCL_PROF_GEMM = 0, TCL_PROF_MATCH = 1, TCL_PROF_NCLS = 2 };

struct TclProfClass {
    bool on = false;
    std::deque<hipEvent_t> ev;            // e0, e1, e0, e1, ... oldest first
    std::vector<hipEvent_t> pool;
    double ms = 0.0, work = 0.0;
    long launches = 0;
    void drain(bool all) {
        while (ev.size() >= 2) {
            hipEvent_t e0 = ev[0], e1 = ev[1];
            if (all) (void)hipEventSynchronize(e1);
            else if (hipEventQuery(e1) != hipSuccess) break;
            float t = 0.f;
            (void)hipEventElapsedTime(&t, e0, e1);
            ms += t;
            pool.push_back(e0); pool.push_back(e1);
            ev.pop_front(); ev.pop_front();
        }
    }
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
};
extern TclProfClass g_tcl_prof[TCL_PROF_NCLS];

// (One host thread issues the launches of both streams -- the Python driver -- so the registry needs no lock; a scope must be closed before the next
// one of its class opens: the deque pairs e0 / e1 by position.)
struct TclProfScope {
    TclProfClass* c; hipStream_t st; hipEvent_t e1; double w;
    TclProfScope(int cls, hipStream_t s, double work) : c(g_tcl_prof[cls].on ? &g_tcl_prof[cls] : nullptr), st(s), e1(nullptr), w(work) {
        if (!c) return;
        if (c->ev.size() > 8192) c->drain(false);
        hipEvent_t e0 = c->get(); e1 = c->get();
        (void)hipEventRecord(e0, st);
        c->ev.push_back(e0);
        c->work += work; c->launches++;
    }
    // this call turned out not to be a plain launch (the in-call tile tuner ran: ~40 launches and a host sync inside the bracket): drop it
    void cancel() {
        if (!c) return;
        c->pool.push_back(c->ev.back()); c->ev.pop_back();
        c->pool.push_back(e1);
        c->work -= w; c->launches--;
        c = nullptr;
    }
    ~TclProfScope() { if (c) { (void)hipEventRecord(e1, st); c->ev.push_back(e1); } }
};
