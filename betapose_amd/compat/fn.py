"""``from fn import getTime`` (fn.py:222-227)."""
import time


def getTime(time1=0):
    if not time1:
        return time.time()
    return time.time(), time.time() - time1
