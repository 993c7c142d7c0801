"""Enums of the plugin surface (subset of srl/base/define.py used on the hot path)."""
import enum


class DoneTypes(enum.Enum):
    RESET = enum.auto()
    NONE = enum.auto()
    TERMINATED = enum.auto()
    TRUNCATED = enum.auto()


class SpaceTypes(enum.Enum):
    UNKNOWN = 0
    DISCRETE = enum.auto()
    CONTINUOUS = enum.auto()
    GRAY_HW = enum.auto()  # (height, width)
    GRAY_HW1 = enum.auto()  # (height, width, 1)
    RGB = enum.auto()  # (height, width, 3)
    FEATURE_MAP = enum.auto()  # (height, width, ch)
    IMAGE_MAP = enum.auto()  # (height, width, ch) stacked frames

    @staticmethod
    def is_image(t) -> bool:
        return t in (SpaceTypes.GRAY_HW, SpaceTypes.GRAY_HW1, SpaceTypes.RGB, SpaceTypes.FEATURE_MAP, SpaceTypes.IMAGE_MAP)


class RLBaseTypes(enum.Flag):
    NONE = 0
    DISCRETE = enum.auto()
    ARRAY_DISCRETE = enum.auto()
    CONTINUOUS = enum.auto()
    ARRAY_CONTINUOUS = enum.auto()
    NP_ARRAY = enum.auto()
    BOX = enum.auto()
