// timer.h -- host wall-clock stopwatch for the benchmark drivers.
//
// Interface contract (what the reference harness calls on its Timer, utils/timer.h:3-45):
//     Timer t;  t.start();  <launch + sync>;  t.stop();  double s = t.elapsedSeconds();
// Own implementation: monotonic steady_clock (the reference mixes system_clock members with
// high_resolution_clock values and only compiles where the two alias), <chrono> included here, reading
// a running watch is allowed, plus min-tracking helpers used by the min-of-40 protocol.
#pragma once

#include <chrono>
#include <limits>

class Timer
{
    using clock = std::chrono::steady_clock;
    using ns    = std::chrono::nanoseconds;

public:
    void start() { m_t0 = clock::now(); m_live = true; }
    void stop() { m_t1 = clock::now(); m_live = false; }

    // nanoseconds between start() and stop() (or now, while running)
    double elapsedNanoseconds() const
    {
        return (double)std::chrono::duration_cast<ns>((m_live ? clock::now() : m_t1) - m_t0).count();
    }
    double elapsedMilliseconds() const { return 1.0e-6 * elapsedNanoseconds(); }
    double elapsedSeconds() const { return 1.0e-9 * elapsedNanoseconds(); }

    // stop, fold the lap into the running minimum, return the lap in seconds
    double lapMin()
    {
        stop();
        const double s = elapsedSeconds();
        if (s < m_best)
            m_best = s;
        return s;
    }
    double best() const { return m_best; }
    void resetBest() { m_best = std::numeric_limits<double>::max(); }

private:
    clock::time_point m_t0{}, m_t1{};
    double m_best = std::numeric_limits<double>::max();
    bool m_live   = false;
};
