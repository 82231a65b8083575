class _Flags:
    def __call__(self, argv, *a, **k):
        return list(argv)


FLAGS = _Flags()
