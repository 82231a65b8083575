"""Stand-in for the third-party `tensorboardX` package (absent from this image): the HARL runners write their scalars through it."""


class SummaryWriter:
    def __init__(self, *a, **k):
        self.scalars = []

    def add_scalar(self, *a, **k):
        self.scalars.append(a)

    def add_scalars(self, *a, **k):
        self.scalars.append(a)

    def export_scalars_to_json(self, *a, **k):
        pass

    def close(self):
        pass
