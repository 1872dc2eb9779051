"""Error classes of the hot path, same names and hierarchy as the reference's
crisprhawk_error.py (lines 9-205) so callers' ``except`` clauses keep working."""


class CrisprHawkError(Exception):
    pass


class CrisprHawkPamError(CrisprHawkError):
    pass


class CrisprHawkIupacTableError(CrisprHawkError):
    pass


class CrisprHawkHaplotypeError(CrisprHawkError):
    pass


class CrisprHawkGuideError(CrisprHawkError):
    pass


class CrisprHawkScoreError(CrisprHawkError):
    pass


class CrisprHawkAzimuthScoreError(CrisprHawkScoreError):
    pass


class CrisprHawkRs3ScoreError(CrisprHawkScoreError):
    pass


class CrisprHawkCfdScoreError(CrisprHawkScoreError):
    pass


class CrisprHawkDeepCpf1ScoreError(CrisprHawkScoreError):
    pass


class CrisprHawkElevationScoreError(CrisprHawkScoreError):
    pass


class CrisprHawkAnnotationError(CrisprHawkError):
    pass


class CrisprHawkOffTargetsError(CrisprHawkError):
    pass
