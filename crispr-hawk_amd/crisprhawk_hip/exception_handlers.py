"""exception_handler with the reference's contract (exception_handlers.py:28-58): raise
``exception_type`` in debug mode, otherwise print ``ERROR: ...`` to stderr and exit with the
given os.EX_* code."""
import sys
from typing import NoReturn, Optional


def exception_handler(exception_type: type, exception: str, code: int, debug: bool,
                      e: Optional[Exception] = None) -> NoReturn:
    if debug:
        if e:
            raise exception_type(f"\n\n{exception}") from e
        raise exception_type(f"\n\n{exception}")
    sys.stderr.write(f"\n\nERROR: {exception}\n")
    sys.exit(code)
