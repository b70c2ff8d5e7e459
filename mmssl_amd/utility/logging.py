"""Run log with the reference's interface — `Logger(filename, is_debug, path).logging(message)` echoes a
time-stamped line and, unless `is_debug`, appends it to `<path>/<filename>`
(/root/reference/MMSSL/utility/logging.py:4-14). Built on the standard library's handlers; the
directory comes from the argument, then $MMSSL_LOG_DIR, then ./logs/ (the reference hard-codes a home
directory) and is created on first use."""
import logging as _pylog
import os
import sys

_STAMP = "%Y-%m-%d %H:%M:"


class Logger:
    def __init__(self, filename, is_debug, path=None):
        self.filename = str(filename)
        self.path = path if path is not None else os.environ.get("MMSSL_LOG_DIR", "./logs/")
        self.log_ = not is_debug
        self._sink = _pylog.Logger("mmssl." + self.filename, level=_pylog.INFO)     # private, not in the registry
        fmt = _pylog.Formatter("%(asctime)s  %(message)s", datefmt=_STAMP)
        echo = _pylog.StreamHandler(sys.stdout)
        echo.setFormatter(fmt)
        self._sink.addHandler(echo)
        self._file_ready = False
        self._fmt = fmt

    def _attach_file(self):
        os.makedirs(self.path, exist_ok=True)
        handler = _pylog.FileHandler(os.path.join(self.path, self.filename), mode="a", delay=True)
        handler.setFormatter(self._fmt)
        self._sink.addHandler(handler)
        self._file_ready = True

    def logging(self, s):
        if self.log_ and not self._file_ready:
            self._attach_file()
        self._sink.info(str(s))
