"""Wall-clock profiler with the reference's log names (`train a step`, `sample_from_replay_buffer`,
`train`, `get_l_probs`, `get_td_error`; reference `algorithm/utils/elapse_timer.py:9-103`,
hooks listed in SURVEY.md §5).  The device work of this build is asynchronous (one graph replay per
step), so these timers measure host enqueue time exactly like the reference's un-synchronised
timers did; device time is measured with HIP events in `bench.py`.
"""
import logging
import time

__all__ = ['UnifiedElapsedTimer', 'ElapsedTimer', 'unified_elapsed_timer']


class ElapsedTimer:
    def __init__(self, log, logger=None, logger_level=logging.DEBUG, repeat=1, force_report=True):
        self._log, self._logger, self._level = log, logger, logger_level
        self._repeat = max(1, int(repeat))
        self._force_report = force_report
        self._n, self._mean, self._last_reported = 0, 0.0, -1.0
        self._skip = False
        self._enabled = (log is not None and logger is not None
                         and logger.getEffectiveLevel() <= logger_level)

    def __enter__(self):
        self._t0 = time.time()
        return self

    def __exit__(self, exc_type, exc, tb):
        if not self._enabled:
            return
        if self._skip:
            self._skip = False
            return
        dt = time.time() - self._t0
        self._n += 1
        self._mean += (dt - self._mean) / self._n
        if self._n % self._repeat == 0:
            if self._force_report or abs(self._mean - self._last_reported) > 0.1:
                self._logger.log(self._level, f'{self._log}: {self._mean:.4f}s')
            self._last_reported = self._mean
            self._n, self._mean = 0, 0.0

    def ignore(self):
        self._skip = True


class UnifiedElapsedTimer:
    def __init__(self, logger=None, logger_level=logging.DEBUG):
        self._logger = logging.getLogger((logger.name if logger is not None else 'asac') + '.profiler')
        self._level = logger_level
        self._timers = {}

    def __call__(self, log, repeat=1, force_report=True) -> ElapsedTimer:
        t = self._timers.get(log)
        if t is None:
            t = self._timers[log] = ElapsedTimer(log, self._logger, self._level, repeat, force_report)
        return t


def unified_elapsed_timer(log, repeat=1, force_report=True, profiler='_profiler'):
    def deco(fn):
        def wrapped(self, *a, **k):
            with getattr(self, profiler)(log, repeat, force_report):
                return fn(self, *a, **k)
        wrapped.__name__ = getattr(fn, '__name__', 'wrapped')
        wrapped.__doc__ = fn.__doc__
        wrapped.__wrapped__ = fn
        wrapped.elapsed_log = log      # (which log line this method reports under: tests/test_host_logic.py)
        return wrapped
    return deco
