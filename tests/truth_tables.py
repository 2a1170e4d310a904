"""Truth tables of the factor functions as pinned by the reference's
test/factor_test.cc:20-236 (arity 1, 2 and the 3-variable IMPLY cases).  Each entry:
(func id, satisfied-bits per position, expected sign)."""
import math

AND, OR, EQUAL, IMPLY_NATURAL, LINEAR, RATIO, LOGICAL, IMPLY_MLN, ISTRUE, AND_CAT = \
    2, 1, 3, 0, 7, 8, 9, 13, 4, 12

CASES = []


def _add(sat, **expect):
    for name, val in expect.items():
        CASES.append((globals()[name], tuple(sat), float(val)))


# ONE_VAR_FACTORS (test/factor_test.cc:27-59)
_add([1], AND=1, OR=1, EQUAL=1, IMPLY_NATURAL=1, LINEAR=1, RATIO=1, LOGICAL=1, ISTRUE=1)
_add([0], AND=-1, OR=-1, EQUAL=1, IMPLY_NATURAL=-1, LINEAR=0, RATIO=0, LOGICAL=0, ISTRUE=-1)
# TWO_VAR_FACTORS (:62-154)
_add([1, 1], AND=1, OR=1, EQUAL=1, IMPLY_NATURAL=1, LINEAR=1, RATIO=1, LOGICAL=1)
_add([1, 0], AND=-1, OR=1, EQUAL=-1, IMPLY_NATURAL=-1, LINEAR=0, RATIO=0, LOGICAL=0)
_add([0, 1], AND=-1, OR=1, EQUAL=-1, IMPLY_NATURAL=0, LINEAR=1, RATIO=1, LOGICAL=1)
_add([0, 0], AND=-1, OR=-1, EQUAL=1, IMPLY_NATURAL=0, LINEAR=1, RATIO=1, LOGICAL=1)
# THREE_VAR_IMPLY (:157-236)
_add([1, 0, 1], IMPLY_NATURAL=0, IMPLY_MLN=1, LINEAR=2, RATIO=math.log2(3.0), LOGICAL=1)
_add([1, 1, 1], IMPLY_NATURAL=1, IMPLY_MLN=1, LINEAR=2, RATIO=math.log2(3.0), LOGICAL=1)
_add([1, 1, 0], IMPLY_NATURAL=-1, IMPLY_MLN=0, LINEAR=0, RATIO=0, LOGICAL=0)
# AND_CATEGORICAL returns 0, not -1, when unsatisfied (src/factor.h:148-158)
_add([1], AND_CAT=1)
_add([0], AND_CAT=0)
_add([1, 0], AND_CAT=0)
_add([1, 1], AND_CAT=1)
