"""Exception names of the reference API (src/gym_duckietown/exceptions.py:10-15)."""
__all__ = ["GymDuckietownException", "InvalidMapException", "NotInLane"]

from dtsim.maps import InvalidMapException as _InvalidMap


class GymDuckietownException(Exception):
    def __init__(self, msg="", **kwargs):
        super().__init__(msg)
        self.info = kwargs


class InvalidMapException(GymDuckietownException, _InvalidMap):
    pass


class NotInLane(GymDuckietownException):
    """Raised when the Duckiebot is not in a lane."""
