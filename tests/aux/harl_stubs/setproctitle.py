"""Stand-in for the third-party `setproctitle` package (absent from this image): the HARL runners call it once at start-up."""


def setproctitle(title):
    pass
