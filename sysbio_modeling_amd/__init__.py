"""sysbio_modeling_amd -- MI355X-native parameter-fitting inner loop behind the
OdeModel / Experiment / TimecourseMeasurement / Project surface of
FedericoV/SysBio_Modeling."""
__version__ = "0.1.0"
