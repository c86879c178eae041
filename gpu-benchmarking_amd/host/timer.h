// timer.h -- host wall-clock stopwatch with the interface the reference harness uses
// (utils/timer.h:3-45: start() / stop() / elapsedNanoseconds() / elapsedSeconds()).
// Written for this project: a monotonic clock (the reference mixes system_clock and
// high_resolution_clock, which only compiles where they alias) and <chrono> included here.
#pragma once

#include <chrono>

class Timer
{
public:
    void start()
    {
        m_begin   = clock::now();
        m_running = true;
    }

    void stop()
    {
        m_end     = clock::now();
        m_running = false;
    }

    double elapsedNanoseconds() const
    {
        const clock::time_point end = m_running ? clock::now() : m_end;
        return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(end - m_begin).count();
    }

    double elapsedSeconds() const
    {
        return elapsedNanoseconds() * 1.0e-9;
    }

private:
    using clock = std::chrono::steady_clock;
    clock::time_point m_begin{}, m_end{};
    bool m_running = false;
};
