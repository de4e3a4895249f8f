"""Device identifier type (mirrors gym_d2d/id.py): a str subclass so ids work as dict keys and in 'tx:rx' joins."""


class Id(str):
    __slots__ = ()
