def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a MI355X (run through gpurun)')
