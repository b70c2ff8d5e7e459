"""`Logger.logging(s)`: print, and append to <path>/<filename> unless is_debug
(interface of /root/reference/MMSSL/utility/logging.py:4-14; the default directory is
configurable here instead of a hard-coded home directory)."""
import os
from datetime import datetime


class Logger:
    def __init__(self, filename, is_debug, path=None):
        self.filename = filename
        self.path = path or os.environ.get("MMSSL_LOG_DIR", "./logs/")
        self.log_ = not is_debug

    def logging(self, s):
        s = str(s)
        stamp = datetime.now().strftime("%Y-%m-%d %H:%M: ")
        print(stamp, s)
        if self.log_:
            os.makedirs(self.path, exist_ok=True)
            with open(os.path.join(self.path, self.filename), "a+") as f:
                f.write(stamp + " " + s + "\n")
