from .homophily_controller import HomophilyMAC

REGISTRY = {"homophily_mac": HomophilyMAC}
