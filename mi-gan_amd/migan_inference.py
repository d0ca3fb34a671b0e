class Generator:  # placeholder
    pass
